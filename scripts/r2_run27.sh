#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r3y}
for n in 2048 16384; do
timeout 1200 python3 bench.py --steps 1 --warmup 0 --pairs 65536 --cpu-sample 0 --recovery-sample 0 --recruit-sample 0 --map-sample 0 --many-alleles-sample 0 --ont-sample $n > gpurun_out/${V}_ont_$n.json 2> gpurun_out/${V}_ont_$n.err
python3 -c "
import json; d=json.load(open('gpurun_out/${V}_ont_$n.json')); print($n, d.get('long_reads'))"
done
