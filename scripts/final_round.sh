#!/bin/bash
# one gpurun call: the bench line, its rocprofv3 kernel stats, and the N = 2 launch path on the one GPU of the box
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-v18}
python3 bench.py > gpurun_out/${V}_bench.json 2> gpurun_out/${V}_bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${V} -o ${V} --output-format csv -- python3 bench.py --steps 2 --warmup 1 --recovery-sample 0 --recruit-sample 0 --pipeline 0 --cpu-sample 0 > gpurun_out/${V}_prof_bench.log 2>&1
timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 1 --warmup 1 --pairs 131072 > gpurun_out/${V}_bench_n2_one_gpu.json 2> gpurun_out/${V}_bench_n2.err
timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 1 --warmup 1 --pairs 131072 --shard-chains > gpurun_out/${V}_bench_n2_shard_chains.json 2> gpurun_out/${V}_bench_n2_sc.err
python3 bench.py --shard-reads --alleles 4096 --pairs 131072 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/${V}_config5_shard.json 2> gpurun_out/${V}_config5_shard.err
cut -c1-300 gpurun_out/${V}_bench.json; echo
cut -c1-300 gpurun_out/${V}_bench_n2_one_gpu.json; echo; tail -2 gpurun_out/${V}_bench_n2.err
cut -c1-300 gpurun_out/${V}_bench_n2_shard_chains.json; echo; tail -2 gpurun_out/${V}_bench_n2_sc.err
grep -o '"ms_per_step": [0-9.]*' gpurun_out/${V}_config5_shard.json
