#!/bin/bash
# PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate runs) of the bench command and of the Gram prefilter probe; SQ counters of the greedy loop
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r3r}
B="bench.py --steps 1 --warmup 1 --cpu-sample 0 --recovery-sample 0 --recruit-sample 0 --ont-sample 0 --map-sample 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch_$V -o f --output-format csv -- python3 $B > gpurun_out/pmc_fetch_$V.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write_$V -o w --output-format csv -- python3 $B > gpurun_out/pmc_write_$V.log 2>&1
F=$(find gpurun_out/pmc_fetch_$V -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/pmc_write_$V -name "*counter_collection.csv" | head -1)
python3 scripts/pmc_summary.py $F $W gpurun_out/${V}_pmc_traffic.json "bench.py --steps 1 --warmup 1 --cpu-sample 0 (two loci of 1M pairs x 256 alleles in the queue, config 2; counted records), MI355X, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (r02 v5 kernels); full-size launches only" 1000000 256
G="scripts/gram_probe.py 131072 4096"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_gfetch_$V -o f --output-format csv -- python3 $G > gpurun_out/pmc_gfetch_$V.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_gwrite_$V -o w --output-format csv -- python3 $G > gpurun_out/pmc_gwrite_$V.log 2>&1
F=$(find gpurun_out/pmc_gfetch_$V -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/pmc_gwrite_$V -name "*counter_collection.csv" | head -1)
python3 scripts/pmc_summary.py $F $W gpurun_out/${V}_pmc_traffic_gram.json "scripts/gram_probe.py 131072 4096 (configs[4] shard shape), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes" 131072 4096 | grep -E "gram|prefilter"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace -d gpurun_out/pmc_sq_$V -o sq --output-format csv -- python3 scripts/solve_scale.py 256 1000000 5000 g 0 > gpurun_out/pmc_sq_$V.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace -d gpurun_out/pmc_mfma_$V -o m --output-format csv -- python3 $G > gpurun_out/pmc_mfma_$V.log 2>&1
python3 - <<PY
import csv, glob, collections
for tag, pat in (("sq","greedy"),("mfma","gram_mfma")):
    fs = glob.glob(f"gpurun_out/pmc_{tag}_${V}/**/*counter_collection.csv", recursive=True)
    if not fs: print(tag, "no csv"); continue
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set); grid={}
    rows=list(csv.DictReader(open(fs[0])))
    big = collections.defaultdict(float)
    for r in rows:
        if pat in r["Kernel_Name"]: big[r["Kernel_Name"]] = max(big[r["Kernel_Name"]], float(r["Grid_Size"]))
    for r in rows:
        k = r["Kernel_Name"]
        if pat not in k or float(r["Grid_Size"]) < 0.5 * big[k]: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in tot:
        print(tag, k.split("(")[0][-40:], "launches", len(n[k]), {c: f"{v/len(n[k]):.4g}" for c, v in tot[k].items()})
PY
tail -2 gpurun_out/pmc_mfma_$V.log
