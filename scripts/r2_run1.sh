#!/bin/bash
# round 2, call 1: the GPU tests, smoke, the bench line with the threaded CPU baseline; host facts of the box
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r2a}
(nproc; grep -m1 "model name" /proc/cpuinfo; grep MemTotal /proc/meminfo; rocm-smi --showmeminfo vram 2>/dev/null | head -8) > gpurun_out/${V}_host.txt 2>&1
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/${V}_pytest.log 2> gpurun_out/${V}_pytest.err
tail -3 gpurun_out/${V}_pytest.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${V}_smoke.log 2>&1; tail -2 gpurun_out/${V}_smoke.log
timeout 1200 python3 bench.py --recovery-sample 0 --recruit-sample 0 --pipeline 0 > gpurun_out/${V}_bench.json 2> gpurun_out/${V}_bench.err
tail -3 gpurun_out/${V}_bench.err
cut -c1-600 gpurun_out/${V}_bench.json
cat gpurun_out/${V}_host.txt
