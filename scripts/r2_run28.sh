#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r3z}
timeout 1200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --pairs 262144 --cpu-sample 0 --recovery-sample 0 --recruit-sample 0 --map-sample 0 --many-alleles-sample 0 --ont-sample 0 > gpurun_out/${V}_n2.json 2> gpurun_out/${V}_n2.err
tail -3 gpurun_out/${V}_n2.err; python3 -c "
import json; d=json.load(open('gpurun_out/${V}_n2.json')); print({k:d[k] for k in ('value','n_gpus','ms_per_step','scaling','steps')}); print(d['config'])"
