#!/bin/bash
# round 2: GPU tests (solver first), smoke, the bench line, rocprofv3 kernel stats of the same command
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r2b}
timeout 900 python3 -m pytest tests/test_gpu_solve.py -x -q > gpurun_out/${V}_pytest_solve.log 2>&1
grep -E "passed|failed|error" gpurun_out/${V}_pytest_solve.log | tail -3
if grep -q "failed\|error" gpurun_out/${V}_pytest_solve.log; then grep -B30 "short test summary" gpurun_out/${V}_pytest_solve.log | tail -60; fi
timeout 1500 python3 -m pytest tests -m gpu -q --deselect tests/test_gpu_solve.py > gpurun_out/${V}_pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/${V}_pytest.log | tail -3
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${V}_smoke.log 2>&1; tail -2 gpurun_out/${V}_smoke.log
timeout 1200 python3 bench.py --recovery-sample 0 --recruit-sample 0 ${BENCH_ARGS} > gpurun_out/${V}_bench.json 2> gpurun_out/${V}_bench.err
tail -3 gpurun_out/${V}_bench.err
python3 - <<PY
import json
try:
    d=json.load(open("gpurun_out/${V}_bench.json"))
    print({k:d[k] for k in ("value","ms_per_step","called_genotype","true_genotype")})
    print(d["kernel_ms_per_step"]); print(d["solver"]["call_by_call_stage_ms"], d["solver"]["all_calls_equal_truth"])
    print(d["roofline"]); print(d.get("vs_cpu_baseline"))
except Exception as e: print("bench json:", e)
PY
if [ "${PROF:-1}" = "1" ]; then
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${V} -o ${V} --output-format csv -- python3 bench.py --steps 2 --warmup 2 --recovery-sample 0 --recruit-sample 0 --cpu-sample 0 > gpurun_out/${V}_prof_bench.log 2>&1
head -12 gpurun_out/prof_${V}/*kernel_stats.csv 2>/dev/null | cut -c1-200
fi
