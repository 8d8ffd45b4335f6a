#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r14}
timeout 1800 python3 -m pytest tests -m gpu -q > gpurun_out/${V}_pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/${V}_pytest.log | tail -2; grep -E "^FAILED|^ERROR|^E  " gpurun_out/${V}_pytest.log | head -20
OFF="--cpu-sample 0 --recovery-sample 0 --recruit-sample 0 --many-alleles-sample 0 --map-sample 0 --ont-sample 0"
timeout 900 python3 bench.py --steps 4 --warmup 2 $OFF > gpurun_out/${V}_bench.json 2> gpurun_out/${V}_bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/${V}_bench.json')); print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'])"
