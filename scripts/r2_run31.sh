#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "== shipped"; timeout 600 python3 scripts/solve_scale.py 256 1000000 256,5000 g 0 2>&1 | grep -E "kind=|lcty solve"
echo "== experiment 8: alternatives beyond the first never evaluated"
LCTY_EXPERIMENT_LIB=$R/locityper_amd/exp/liblocityper_hip_exp8.so timeout 600 python3 scripts/solve_scale.py 256 1000000 256,5000 g 0 2>&1 | grep -E "kind=|lcty solve"
