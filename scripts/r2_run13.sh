#!/bin/bash
# what a random 32-byte gather costs: footprint x wavefronts x gathers in flight (scripts/gather_probe.hip)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r2n}
hipcc -O3 --offload-arch=gfx950 -o /tmp/gather_probe scripts/gather_probe.hip
P=/tmp/gather_probe
{
for gb in 1 8 40 148; do
  timeout 120 $P $gb 64 40 1 20000
  timeout 120 $P $gb 64 40 4 20000
done
for gb in 8 148; do
  for w in 1250 2500 5000 10000; do
    timeout 120 $P $gb $w 40 1 20000
    timeout 120 $P $gb $w 40 4 10000
  done
  timeout 120 $P $gb 1250 40 1 20000 32
  timeout 120 $P $gb 1250 64 1 20000
  timeout 120 $P $gb 1250 10 1 20000
  timeout 120 $P $gb 20000 64 8 5000
done
} > gpurun_out/${V}_gather.log 2>&1
cat gpurun_out/${V}_gather.log
