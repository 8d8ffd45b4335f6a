#!/bin/bash
# Counter passes over a probe script (each pass alone, --kernel-trace only):  scripts/pmc_probe.sh <tag> <kernel substring> -- python3 scripts/<probe>.py args...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
TAG=$1; KERN=$2; shift; shift; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace -d "$OUT/p1" -o a --output-format csv -- "$@" > "$OUT/p1.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_INSTS_FLAT --kernel-trace -d "$OUT/p2" -o b --output-format csv -- "$@" > "$OUT/p2.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace -d "$OUT/p3" -o c --output-format csv -- "$@" > "$OUT/p3.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/p4" -o d --output-format csv -- "$@" > "$OUT/p4.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/p5" -o e --output-format csv -- "$@" > "$OUT/p5.log" 2>&1
python3 - "$OUT" "$KERN" <<'PY'
import csv, glob, sys, collections, json
out, kern = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        if kern not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        seen.add((r["Counter_Name"], r["Dispatch_Id"]))
    for c, d in seen: n[c] += 1
res = {c: {"sum": tot[c], "dispatches": n[c]} for c in sorted(tot)}
json.dump({"kernel_contains": kern, "counters": res}, open(out + "/summary.json", "w"), indent=1)
for c in sorted(tot): print(f"{c:34s} {tot[c]:.6g}  ({n[c]} dispatches)")
PY
