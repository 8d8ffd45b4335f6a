#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r2c}
export LIBC_FATAL_STDERR_=1 AMD_LOG_LEVEL=1
timeout 900 python3 -X faulthandler -m pytest tests/test_gpu_solve.py -x -q > gpurun_out/${V}_pytest_solve.log 2>&1
grep -E "passed|failed|error" gpurun_out/${V}_pytest_solve.log | tail -3
grep -v "^  File\|^$" gpurun_out/${V}_pytest_solve.log | head -40
unset AMD_LOG_LEVEL
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_streaming.py tests/test_gpu_pins.py tests/test_gpu_debug_dumps.py tests/test_gpu_host_api.py -q -k "adversarial or chain_sharded or streaming_equals or pins or debug or host" > gpurun_out/${V}_pytest_some.log 2>&1
grep -E "passed|failed|error" gpurun_out/${V}_pytest_some.log | tail -3
grep -E "^FAILED|^ERROR|^E  " gpurun_out/${V}_pytest_some.log | head -30
timeout 1200 python3 bench.py --recovery-sample 0 --recruit-sample 0 --cpu-sample 0 ${BENCH_ARGS} > gpurun_out/${V}_bench.json 2> gpurun_out/${V}_bench.err
tail -3 gpurun_out/${V}_bench.err
python3 - <<PY
import json
try:
    d=json.load(open("gpurun_out/${V}_bench.json"))
    print({k:d[k] for k in ("value","ms_per_step","called_genotype","true_genotype")})
    print(d["kernel_ms_per_step"]); print(d["solver"]["call_by_call_stage_ms"], d["solver"]["all_calls_equal_truth"])
    print(d["roofline"])
except Exception as e: print("bench json:", e)
PY
