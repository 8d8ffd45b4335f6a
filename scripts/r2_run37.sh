#!/bin/bash
# PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate runs) of the bench command, final kernels of the round
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r02v8}
B="bench.py --steps 1 --warmup 1 --cpu-sample 0 --recovery-sample 0 --recruit-sample 0 --ont-sample 0 --map-sample 0 --many-alleles-sample 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch_$V -o f --output-format csv -- python3 $B > gpurun_out/pmc_fetch_$V.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write_$V -o w --output-format csv -- python3 $B > gpurun_out/pmc_write_$V.log 2>&1
F=$(find gpurun_out/pmc_fetch_$V -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/pmc_write_$V -name "*counter_collection.csv" | head -1)
python3 scripts/pmc_summary.py $F $W gpurun_out/${V}_pmc_traffic.json "bench.py --steps 1 --warmup 1 --cpu-sample 0 (two loci of 1M pairs x 256 alleles in the queue, config 2; counted records), MI355X, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (r02 v8 kernels); full-size launches only" 1000000 256
