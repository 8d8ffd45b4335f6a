#!/bin/bash
# one gpurun call: the two PMC passes of the bench command (never combined with other traces), summarised into profiles/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-v19}
B="bench.py --steps 1 --warmup 0 --cpu-sample 0 --recovery-sample 0 --recruit-sample 0 --pipeline 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch_$V -o f --output-format csv -- python3 $B > gpurun_out/pmc_fetch_$V.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write_$V -o w --output-format csv -- python3 $B > gpurun_out/pmc_write_$V.log 2>&1
F=$(find gpurun_out/pmc_fetch_$V -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/pmc_write_$V -name "*counter_collection.csv" | head -1)
python3 scripts/pmc_summary.py $F $W gpurun_out/${V}_pmc_traffic.json "bench.py --steps 1 --warmup 0 --cpu-sample 0 (1M pairs x 256 alleles, config 2), MI355X, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes ($V kernels)" 1000000 256
