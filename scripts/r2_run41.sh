#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r02v9}
S=$(date +%s)
timeout 1700 python3 bench.py > gpurun_out/${V}_bench.json 2> gpurun_out/${V}_bench.err
echo "bench wall $(( $(date +%s) - S )) s"; tail -2 gpurun_out/${V}_bench.err
python3 - <<PY
import json
d=json.load(open("gpurun_out/${V}_bench.json"))
print({k:d[k] for k in ("value","ms_per_step","steps","called_genotype","true_genotype")})
print(d["kernel_ms_per_step"]); print(d["roofline"]); print(d.get("vs_cpu_baseline"))
for k in ("candidate_generation","long_reads","recovery","recruitment","many_alleles"):
    print(k, {a:b for a,b in d[k].items() if a not in ("sample","workload")})
PY
timeout 300 python3 -c "
import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
