#!/bin/bash
# SQ and instruction-cache counters of alignment recovery's transfer_kernel (scripts/ont_recover_probe.py under rocprofv3, two passes);
# -> <out>/transfer_sq.json (copy to profiles/r05_pmc_sq_transfer_kernel.json)
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/${1:-transfersq}
READS=${2:-2048}
mkdir -p $OUT
python3 scripts/sources_sha.py > $OUT/sources.sha16
timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --kernel-trace -d $OUT/a -o a --output-format csv -- python3 scripts/ont_recover_probe.py $READS > $OUT/a.log 2>&1; echo "icache rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace -d $OUT/b -o b --output-format csv -- python3 scripts/ont_recover_probe.py $READS > $OUT/b.log 2>&1; echo "sq rc=$?"
python3 - "$OUT" "$READS" <<'PY'
import csv, glob, json, re, sys
out, reads = sys.argv[1], int(sys.argv[2])
tot = {}
for d in ("a", "b"):
    for f in glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "transfer_kernel" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
m = re.search(r"new (\d+) \(", open(out + "/b.log").read())
transfers = 2 * (int(m.group(1)) if m else reads * 255)                       # the probe recovers the batch twice
res = {"reads": reads, "alleles": 256, "transfers": transfers, "counters": tot,
       "per_chunk_of_64_transfers": {k: tot[k] * 64.0 / transfers for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU") if k in tot},
       "issuing_frac": tot.get("SQ_ACTIVE_INST_ANY", 0.0) / max(tot.get("SQ_WAVE_CYCLES", 0.0), 1.0),
       "waiting_frac": tot.get("SQ_WAIT_ANY", 0.0) / max(tot.get("SQ_WAVE_CYCLES", 0.0), 1.0),
       "icache_miss_frac": tot.get("SQC_ICACHE_MISSES", 0.0) / max(tot.get("SQC_ICACHE_REQ", 0.0), 1.0),
       "sources_sha16": open(out + "/sources.sha16").read().strip(),
       "note": "transfer_kernel over scripts/ont_recover_probe.py (10-kb ONT reads, primaries only, 256 alleles), dry pass and walk launches summed"}
json.dump(res, open(out + "/transfer_sq.json", "w"), indent=1)
print(json.dumps(res))
PY
