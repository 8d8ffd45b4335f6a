#!/bin/bash
# HBM traffic of alignment recovery's transfer_kernel: FETCH_SIZE and WRITE_SIZE, each pass alone (scripts/ont_recover_probe.py under rocprofv3);
# bytes per transferred alignment of 10-kb ONT reads on 256 alleles -> <out>/transfer_traffic.json (copy to profiles/r05_pmc_transfer_kernel.json)
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/${1:-transfertraffic}
READS=${2:-2048}
mkdir -p $OUT
python3 scripts/sources_sha.py > $OUT/sources.sha16
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o f --output-format csv -- python3 scripts/ont_recover_probe.py $READS > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o w --output-format csv -- python3 scripts/ont_recover_probe.py $READS > $OUT/write.log 2>&1; echo "write rc=$?"
python3 - "$OUT" "$READS" <<'PY'
import csv, glob, sys, json, re
out, reads = sys.argv[1], int(sys.argv[2])
tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}; launches = {"FETCH_SIZE": 0, "WRITE_SIZE": 0}
for d, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for f in glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            if "transfer_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name:
                tot[name] += float(r["Counter_Value"]); seen.add(r["Dispatch_Id"])
        launches[name] += len(seen)
# the probe scores + recovers the batch twice (once to size the context's scratch, once timed): transfers = reads x 255 targets per run
log = open(out + "/fetch.log").read()
m = re.search(r"new (\d+) \(", log)
runs = 2
transfers = runs * (int(m.group(1)) if m else reads * 255)
res = {"reads": reads, "alleles": 256, "runs_of_the_probe": runs, "transfers": transfers, "launches": launches,
       "fetch_bytes_raw": 1024.0 * tot["FETCH_SIZE"], "write_bytes": 1024.0 * tot["WRITE_SIZE"],
       "bytes_per_transfer": {"fetch_raw": 1024.0 * tot["FETCH_SIZE"] / transfers, "write": 1024.0 * tot["WRITE_SIZE"] / transfers},
       "sources_sha16": open(out + "/sources.sha16").read().strip(),
       "note": "transfer_kernel over scripts/ont_recover_probe.py (10-kb ONT reads, primaries only, 256 alleles, transfer_fails 100), dry pass and walk launches summed; FETCH_SIZE raw (narrow gathers: not doubled)"}
json.dump(res, open(out + "/transfer_traffic.json", "w"), indent=1)
print(json.dumps(res))
PY
