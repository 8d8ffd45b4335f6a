#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r2o}
{
echo "== shipped"; timeout 600 python3 scripts/solve_scale.py 256 1000000 256,5000 g 4 2>&1 | grep -E "kind=|lcty solve"
for N in 1 2 3 4 7; do
  echo "== experiment $N"
  LCTY_EXPERIMENT_LIB=$R/locityper_amd/exp/liblocityper_hip_exp$N.so timeout 600 python3 scripts/solve_scale.py 256 1000000 256,5000 g 4 2>&1 | grep -E "kind=|lcty solve"
done
} > gpurun_out/${V}_exp.log 2>&1
cat gpurun_out/${V}_exp.log
