#!/bin/bash
# the queue with other chains-per-wavefront of the greedy loop / other LDS forms of the annealing chains
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r19}
OFF="--cpu-sample 0 --recovery-sample 0 --recruit-sample 0 --many-alleles-sample 0 --map-sample 0 --ont-sample 0"
for KN in "--knob solve_chains_per_wave=6 --knob anneal_lds_weights=0" "--knob anneal_lds_weights=0" "--knob solve_chains_per_wave=6 --knob anneal_lds_weights=1"; do
timeout 900 python3 bench.py --steps 10 --warmup 2 $OFF $KN > gpurun_out/${V}_x.json 2> gpurun_out/${V}_x.err
python3 -c "
import json; d=json.load(open('gpurun_out/${V}_x.json')); print('$KN', d['ms_per_step'], d['kernel_ms_per_step'])"
done
