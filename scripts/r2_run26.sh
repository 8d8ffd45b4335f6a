#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r3w}
timeout 900 python3 bench.py --shard-reads --alleles 4096 --pairs 131072 --steps 2 --warmup 1 --cpu-sample 0 --recovery-sample 0 --recruit-sample 0 --ont-sample 0 --map-sample 0 --many-alleles-sample 0 > gpurun_out/${V}_config5_shard.json 2> gpurun_out/${V}_config5_shard.err
tail -2 gpurun_out/${V}_config5_shard.err; python3 -c "
import json; d=json.load(open('gpurun_out/${V}_config5_shard.json')); print({k:d[k] for k in ('value','ms_per_step','config')}); print(d.get('kernel_ms_per_step'))"
timeout 900 python3 scripts/ont_stream_probe.py 16384 256 4096 > gpurun_out/${V}_ont_pageable.json 2> gpurun_out/${V}_ont.err
timeout 900 python3 scripts/ont_stream_probe.py 16384 256 4096 pinned > gpurun_out/${V}_ont_pinned.json 2>> gpurun_out/${V}_ont.err
python3 -c "
import json
for n in ('pageable','pinned'):
    d=json.load(open('gpurun_out/${V}_ont_'+n+'.json')); print(n, {k:d[k] for k in ('upload_s','upload_GBs','raw_GB','score_kernel_ms_total','reads_per_s_kernel','called','true')})"
tail -2 gpurun_out/${V}_ont.err
