#!/bin/bash
# SQ / TCC counters of the greedy loop (5 000 chains, config 2): separate --pmc passes, --kernel-trace only
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r2h}
CMD="scripts/solve_scale.py 256 1000000 5000 g 4"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace -d gpurun_out/pmc_sq_$V -o sq --output-format csv -- python3 $CMD > gpurun_out/pmc_sq_$V.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace -d gpurun_out/pmc_lds_$V -o lds --output-format csv -- python3 $CMD > gpurun_out/pmc_lds_$V.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace -d gpurun_out/pmc_tcc_$V -o tcc --output-format csv -- python3 $CMD > gpurun_out/pmc_tcc_$V.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch_$V -o f --output-format csv -- python3 $CMD > gpurun_out/pmc_fetch_$V.log 2>&1
python3 - <<PY
import csv, glob, collections
for tag in ("sq","lds","tcc","fetch"):
    fs = glob.glob(f"gpurun_out/pmc_{tag}_${V}/**/*counter_collection.csv", recursive=True)
    if not fs: print(tag, "no csv"); continue
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][-40:]
        if "greedy" not in k and "solve_init" not in k: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in tot:
        print(tag, k, "launches", len(n[k]), {c: f"{v/len(n[k]):.4g}" for c, v in tot[k].items()})
PY
tail -2 gpurun_out/pmc_sq_$V.log
