#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r2e}
timeout 300 python3 .scratch/repro_abort.py > gpurun_out/${V}_repro.log 2>&1; echo "repro rc=$?"; tail -4 gpurun_out/${V}_repro.log
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/${V}_pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/${V}_pytest.log | tail -3
grep -E "^FAILED|^ERROR|^E  " gpurun_out/${V}_pytest.log | head -30
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${V}_smoke.log 2>&1; tail -1 gpurun_out/${V}_smoke.log
timeout 1200 python3 bench.py --recovery-sample 0 --recruit-sample 0 --cpu-sample 0 ${BENCH_ARGS} > gpurun_out/${V}_bench.json 2> gpurun_out/${V}_bench.err
tail -3 gpurun_out/${V}_bench.err
python3 - <<PY
import json
try:
    d=json.load(open("gpurun_out/${V}_bench.json"))
    print({k:d[k] for k in ("value","ms_per_step","called_genotype","true_genotype")})
    print(d["kernel_ms_per_step"]); print(d["solver"]["call_by_call_stage_ms"], d["solver"]["all_calls_equal_truth"])
    print({k:(v["launch_ms"], round(v["frac"],4)) for k,v in d["roofline_all"].items()})
except Exception as e: print("bench json:", e)
PY
