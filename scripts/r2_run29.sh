#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r4a}
timeout 1200 python3 -m pytest tests/test_gpu_solve.py -m gpu -q -x > gpurun_out/${V}_pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/${V}_pytest.log | tail -2; grep -E "^FAILED|^ERROR|^E  " gpurun_out/${V}_pytest.log | head -30
timeout 600 python3 scripts/solve_scale.py 256 1000000 400 a 0 2>&1 | grep -E "kind=" > gpurun_out/${V}_scale.log
cat gpurun_out/${V}_scale.log
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${V} -o ${V} --output-format csv -- python3 bench.py --steps 6 --warmup 2 --recovery-sample 0 --recruit-sample 0 --ont-sample 0 --cpu-sample 0 --map-sample 0 --many-alleles-sample 0 > gpurun_out/${V}_prof_bench.json 2> gpurun_out/${V}_prof_bench.err
python3 - <<PY
import json
try:
    d=json.load(open("gpurun_out/${V}_prof_bench.json"))
    print({k:d[k] for k in ("value","ms_per_step","steps","called_genotype","true_genotype")})
    print(d["kernel_ms_per_step"])
except Exception as e: print("bench json:", e)
PY
head -4 gpurun_out/prof_${V}/*kernel_stats.csv 2>/dev/null | cut -c1-160
