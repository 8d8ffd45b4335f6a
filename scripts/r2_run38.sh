#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r13}
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -E "passed|failed|^E " | tail -8
timeout 900 python3 bench.py --shard-reads --pairs 131072 --alleles 4096 --steps 5 --warmup 2 --cpu-sample 0 > gpurun_out/${V}_cfg5.json 2> gpurun_out/${V}_cfg5.err
tail -2 gpurun_out/${V}_cfg5.err
python3 -c "
import json; d=json.load(open('gpurun_out/${V}_cfg5.json')); print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'])"
