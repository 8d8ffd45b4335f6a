#!/bin/bash
# Memory-system and issue counters of one kernel of a probe script (each pass alone, --kernel-trace only):
#   scripts/pmc_mem.sh <tag> <kernel substring> -- python3 scripts/<probe>.py args...
# Prints, per counter, the value of the LARGEST dispatch of the kernel (by grid) — the full-size launch of the probe.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
TAG=$1; KERN=$2; shift; shift; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
pass() { n=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d "$OUT/$n" -o $n --output-format csv -- "${CMD[@]}" > "$OUT/$n.log" 2>&1; echo "pass $n rc=$?"; }
CMD=("$@")
pass m1 TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE
pass m2 TCC_READ_sum TCC_READ_SECTORS_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass m3 SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
pass m4 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES
pass m5 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TA_FLAT_READ_WAVEFRONTS_sum
python3 - "$OUT" "$KERN" <<'PY'
import csv, glob, sys, collections, json
out, kern = sys.argv[1], sys.argv[2]
res = {}
for f in sorted(glob.glob(out + "/m*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float)); grid = {}
    for r in csv.DictReader(open(f)):
        if kern not in r["Kernel_Name"]: continue
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        grid[r["Dispatch_Id"]] = float(r.get("Grid_Size") or 0)
    if not grid: continue
    big = max(grid, key=lambda d: (grid[d], int(d)))
    for c, v in per[big].items(): res[c] = v
    res.setdefault("_grid", grid[big])
json.dump({"kernel_contains": kern, "largest_dispatch": res}, open(out + "/mem_summary.json", "w"), indent=1)
for c in sorted(res): print(f"{c:40s} {res[c]:.6g}")
PY
grep -h "init\|loop" "$OUT"/m1.log | tail -4
