#!/bin/bash
# one gpurun call: kernel-trace stats of the bench command + the two PMC passes (never combined with other traces)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_v3 -o v3 --output-format csv -- python3 bench.py --steps 2 --warmup 1 --recovery-sample 0 --recruit-sample 0 --pipeline 0 > gpurun_out/prof_v3_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -o f --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --recovery-sample 0 --recruit-sample 0 --pipeline 0 > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write -o w --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --recovery-sample 0 --recruit-sample 0 --pipeline 0 > gpurun_out/pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_rec -o rec --output-format csv -- python3 scripts/recovery_probe.py 1000000 256 > gpurun_out/prof_rec.log 2>&1
python3 bench.py > gpurun_out/bench_plain.json 2> gpurun_out/bench_plain.err
grep '^{' gpurun_out/prof_v3_bench.log | cut -c1-400
ls gpurun_out/prof_v3 gpurun_out/pmc_fetch gpurun_out/pmc_write
