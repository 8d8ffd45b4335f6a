#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r22}
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_bam.py tests/test_gpu_example.py tests/test_gpu_streaming.py -m gpu -q -x 2>&1 | grep -E "passed|failed|^E " | tail -5
timeout 900 python3 bench.py --steps 1 --warmup 0 --pairs 8192 --cpu-sample 0 --recruit-sample 0 --many-alleles-sample 0 --map-sample 0 --ont-sample 1024 > gpurun_out/${V}_b.json 2> gpurun_out/${V}_b.err
python3 -c "
import json; d=json.load(open('gpurun_out/${V}_b.json')); print('long_reads set_hap_alns_s', d['long_reads']['set_hap_alns_s'], 'recovery set_hap_alns_s', d['recovery']['set_hap_alns_s'], d['recovery']['alignments_transferred'])"
