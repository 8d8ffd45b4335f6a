#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r15}
OFF="--cpu-sample 0 --recovery-sample 0 --recruit-sample 0 --many-alleles-sample 0 --map-sample 0 --ont-sample 0"
timeout 900 python3 bench.py --steps 4 --warmup 2 $OFF --knob queue_trace=1 > gpurun_out/${V}_bench.json 2> gpurun_out/${V}_trace.err
python3 -c "
import json; d=json.load(open('gpurun_out/${V}_bench.json')); print(d['ms_per_step'])"
grep "lcty queue" gpurun_out/${V}_trace.err | tail -60
