# HBM traffic of the long route's kernels: FETCH_SIZE and WRITE_SIZE, each pass alone (scripts/map_long_probe.py under rocprofv3)
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/${1:-longtraffic}
mkdir -p $OUT
ARGS="--reads 2048 --alleles 16 --reps 1"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o f --output-format csv -- python3 scripts/map_long_probe.py $ARGS > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o w --output-format csv -- python3 scripts/map_long_probe.py $ARGS > $OUT/write.log 2>&1; echo "write rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
res = collections.defaultdict(dict)
for d, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for f in glob.glob(sys.argv[1] + "/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "map_long" in r["Kernel_Name"] and r["Counter_Name"] == name:
                k = "align" if "align" in r["Kernel_Name"] else "chain" if "chain" in r["Kernel_Name"] else "emit"
                res[k][name] = res[k].get(name, 0.0) + float(r["Counter_Value"])
print(json.dumps(res))
json.dump(res, open(sys.argv[1] + "/traffic.json", "w"))
PY
