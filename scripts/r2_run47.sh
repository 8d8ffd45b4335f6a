#!/bin/bash
# SQ counters (issue / wait split, instruction mix) of the kernels of the queue, final state of round 2
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r02v10}
B="bench.py --steps 2 --warmup 2 --cpu-sample 0 --recovery-sample 0 --recruit-sample 0 --ont-sample 0 --map-sample 0 --many-alleles-sample 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d gpurun_out/pmc_sq_$V -o sq --output-format csv -- python3 $B > gpurun_out/pmc_sq_$V.log 2>&1
python3 - <<PY
import csv, glob, collections, json
fs = glob.glob("gpurun_out/pmc_sq_$V/**/*counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0])))
big = collections.defaultdict(float)
for r in rows: big[r["Kernel_Name"]] = max(big[r["Kernel_Name"]], float(r["Grid_Size"]))
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"]
    if float(r["Grid_Size"]) < 0.5 * big[k]: continue
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
out = {}
for k in tot:
    if not any(x in k for x in ("greedy_loop", "anneal_loop", "solve_init", "score_counted", "prefilter_tile", "build_loc_table")): continue
    c = {a: v / len(n[k]) for a, v in tot[k].items()}
    c["launches"] = len(n[k])
    if c.get("SQ_WAVE_CYCLES"):
        c["active_over_wave_cycles"] = c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]; c["wait_over_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    out[k.split("(")[0]] = c
json.dump({"note": "rocprofv3 --pmc (one pass) of bench.py --steps 2 --warmup 2 (queue, config 2), full-size launches, per launch; r02 v10 kernels", "kernels": out}, open("gpurun_out/${V}_pmc_sq.json", "w"), indent=1)
for k, c in out.items(): print(k[-40:], {a: (f"{v:.3g}" if isinstance(v, float) else v) for a, v in c.items()})
PY
