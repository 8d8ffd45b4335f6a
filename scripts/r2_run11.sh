#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r2l}
timeout 600 python3 scripts/own_tables_probe.py > gpurun_out/${V}_own_tables.log 2>&1; cat gpurun_out/${V}_own_tables.log | tail -9
timeout 1800 python3 -m pytest tests -m gpu -q > gpurun_out/${V}_pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/${V}_pytest.log | tail -2; grep -E "^FAILED|^ERROR|^E  " gpurun_out/${V}_pytest.log | head -20
timeout 1500 python3 bench.py > gpurun_out/${V}_bench.json 2> gpurun_out/${V}_bench.err
tail -2 gpurun_out/${V}_bench.err
python3 - <<PY
import json
try:
    d=json.load(open("gpurun_out/${V}_bench.json"))
    print({k:d[k] for k in ("value","ms_per_step","steps","called_genotype","true_genotype")})
    print(d["kernel_ms_per_step"]); print(d["solver"]["call_by_call_stage_ms"], d["solver"]["all_calls_equal_truth"])
    print({k:(round(v["launch_ms"],2), round(v["frac"],4)) for k,v in d["roofline_all"].items()}); print(d.get("vs_cpu_baseline"))
except Exception as e: print("bench json:", e)
PY
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${V} -o ${V} --output-format csv -- python3 bench.py --steps 4 --warmup 2 --recovery-sample 0 --recruit-sample 0 --ont-sample 0 --cpu-sample 0 --map-sample 0 --many-alleles-sample 0 > gpurun_out/${V}_prof_bench.log 2>&1
head -8 gpurun_out/prof_${V}/*kernel_stats.csv 2>/dev/null | cut -c1-160
