#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r2j}
timeout 900 python3 -m pytest tests/test_gpu_solve.py tests/test_gpu_counted.py -q -x > gpurun_out/${V}_pytest_a.log 2>&1
grep -E "passed|failed|error" gpurun_out/${V}_pytest_a.log | tail -2; grep -E "^FAILED|^ERROR|^E  " gpurun_out/${V}_pytest_a.log | head -20
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -k "10kb" > gpurun_out/${V}_pytest_b.log 2>&1
grep -E "passed|failed|error" gpurun_out/${V}_pytest_b.log | tail -2; grep -E "^FAILED|^ERROR|^E  " gpurun_out/${V}_pytest_b.log | head -20
timeout 900 python3 scripts/solve_scale.py 256 1000000 5000 g 4,2 > gpurun_out/${V}_scale.log 2>&1; tail -2 gpurun_out/${V}_scale.log
timeout 1500 python3 bench.py --cpu-sample 0 > gpurun_out/${V}_bench.json 2> gpurun_out/${V}_bench.err
tail -3 gpurun_out/${V}_bench.err
python3 - <<PY
import json
try:
    d=json.load(open("gpurun_out/${V}_bench.json"))
    print({k:d[k] for k in ("value","ms_per_step","called_genotype","true_genotype")})
    print(d["kernel_ms_per_step"]); print(d["solver"]["call_by_call_stage_ms"], d["solver"]["all_calls_equal_truth"])
    print({k:(round(v["launch_ms"],2), round(v["frac"],4)) for k,v in d["roofline_all"].items()})
    print("long_reads", d.get("long_reads")); print("recovery", d.get("recovery")); print("recruitment", d.get("recruitment"))
except Exception as e: print("bench json:", e)
PY
