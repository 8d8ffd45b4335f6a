#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r3t}
T0=$(date +%s); python3 bench.py > gpurun_out/${V}_bench.json 2> gpurun_out/${V}_bench.err; echo "bench wall $(( $(date +%s) - T0 )) s"
tail -2 gpurun_out/${V}_bench.err
python3 - <<PY
import json
d=json.load(open("gpurun_out/${V}_bench.json"))
print({k:d[k] for k in ("value","ms_per_step","steps")}); print(d.get("many_alleles")); print(d.get("candidate_generation"))
PY
