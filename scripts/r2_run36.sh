#!/bin/bash
# timing probe: annealing chains of locus i split around the greedy launch of locus i + 1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r12}
OFF="--cpu-sample 0 --recovery-sample 0 --recruit-sample 0 --many-alleles-sample 0 --map-sample 0 --ont-sample 0"
for P in 0 300000 450000 600000; do
  K=""; [ $P -gt 0 ] && K="--knob anneal_split_probe=$P"
  timeout 900 python3 bench.py --steps 4 --warmup 2 $OFF $K > gpurun_out/${V}_probe$P.json 2> gpurun_out/${V}_probe$P.err
  python3 -c "
import json; d=json.load(open('gpurun_out/${V}_probe$P.json')); print($P, d['ms_per_step'], d['kernel_ms_per_step'])"
done
