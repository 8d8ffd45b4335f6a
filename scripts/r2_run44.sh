#!/bin/bash
# rocprofv3 kernel stats of the default bench command (CPU baseline off: it launches nothing)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r02v10}
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${V} -o ${V} --output-format csv -- python3 bench.py --cpu-sample 0 > gpurun_out/${V}_prof_bench.log 2>&1
tail -1 gpurun_out/${V}_prof_bench.log | head -c 300; echo
head -12 gpurun_out/prof_${V}/*kernel_stats.csv 2>/dev/null | cut -c1-170
