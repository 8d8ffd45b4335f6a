#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r3a}
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "matrix_cores or chain_sharded or read_sharded" > gpurun_out/${V}_pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/${V}_pytest.log | tail -2; grep -E "^FAILED|^ERROR|^E  " gpurun_out/${V}_pytest.log | head -30
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${V} -o ${V} --output-format csv -- python3 scripts/gram_probe.py 131072 4096 > gpurun_out/${V}_gram_4096.json 2> gpurun_out/${V}_gram.err
cat gpurun_out/${V}_gram_4096.json
python3 - <<PY
import csv,glob
for r in csv.DictReader(open(glob.glob("gpurun_out/prof_${V}/*kernel_stats.csv")[0])):
    if 'gram' in r['Name'] or 'prefilter' in r['Name']:
        print(r['Name'].split('(')[0][-45:], r['Calls'], round(float(r['AverageNs'])/1e6,3), 'ms')
PY
