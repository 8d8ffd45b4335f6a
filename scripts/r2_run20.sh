#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r2y}
timeout 1500 python3 scripts/gram_probe.py 131072 4096 > gpurun_out/${V}_gram_4096.json 2> gpurun_out/${V}_gram.err; cat gpurun_out/${V}_gram_4096.json; tail -3 gpurun_out/${V}_gram.err
