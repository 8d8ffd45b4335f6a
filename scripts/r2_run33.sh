#!/bin/bash
# SQ counters of the long-read alignment recovery (transfer_kernel) in the bench's long_reads leg
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r7}
cd /tmp && export TMPDIR=/tmp && cd $R
B="bench.py --steps 1 --warmup 0 --pairs 8192 --cpu-sample 0 --recovery-sample 0 --recruit-sample 0 --many-alleles-sample 0 --map-sample 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d gpurun_out/pmc_xfer_$V -o x --output-format csv -- python3 $B > gpurun_out/pmc_xfer_$V.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --kernel-trace -d gpurun_out/pmc_xfer2_$V -o x --output-format csv -- python3 $B > gpurun_out/pmc_xfer2_$V.log 2>&1
python3 - <<PY
import csv, glob, collections
for tag in ("xfer","xfer2"):
    fs = glob.glob(f"gpurun_out/pmc_{tag}_$V/**/*counter_collection.csv", recursive=True)
    if not fs: print(tag, "no csv"); continue
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "transfer_kernel" not in k: continue
        key = (k.split("(")[0][-30:], r["Grid_Size"])
        tot[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[key].add(r["Dispatch_Id"])
    for k in tot: print(tag, k, "launches", len(n[k]), {c: f"{v/len(n[k]):.4g}" for c, v in tot[k].items()})
PY
tail -1 gpurun_out/pmc_xfer_$V.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d.get('long_reads'))"
