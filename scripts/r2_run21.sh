#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r2z}
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${V} -o ${V} --output-format csv -- python3 scripts/gram_probe.py 131072 4096 > gpurun_out/${V}_gram_4096.json 2> gpurun_out/${V}_gram.err
cat gpurun_out/${V}_gram_4096.json
grep -E "gram|prefilter" gpurun_out/prof_${V}/*kernel_stats.csv | cut -c1-200
