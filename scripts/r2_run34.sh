#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r8}
for N in 2048 6144; do
timeout 900 python3 bench.py --steps 1 --warmup 0 --pairs 8192 --cpu-sample 0 --recovery-sample 0 --recruit-sample 0 --many-alleles-sample 0 --map-sample 0 --ont-sample $N > gpurun_out/${V}_ont$N.json 2> gpurun_out/${V}_ont$N.err
tail -3 gpurun_out/${V}_ont$N.err
python3 -c "
import json; d=json.load(open('gpurun_out/${V}_ont$N.json')); print(d.get('long_reads'))"
done
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_streaming.py -m gpu -q -x -k "recover or transfer or long or ont or stream" 2>&1 | tail -3
