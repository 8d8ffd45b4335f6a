#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r2d}
timeout 300 python3 .scratch/repro_abort.py > gpurun_out/${V}_repro.log 2>&1; echo "repro rc=$?"; tail -5 gpurun_out/${V}_repro.log
timeout 600 /opt/rocm/bin/rocgdb -batch -ex run -ex bt -ex "info threads" --args python3 .scratch/repro_abort.py > gpurun_out/${V}_gdb.log 2>&1
grep -v "^\[New Thread\|^\[Thread.*exited\|^warning" gpurun_out/${V}_gdb.log | tail -60
