#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
for pf in 1 0 3 2; do echo "prefetch/exp=$pf"; timeout 600 python3 scripts/solve_scale.py 256 1000000 256,5000 g 4 $pf 2>&1 | tail -2; done
