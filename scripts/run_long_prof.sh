# the long route of candidate generation under rocprofv3: kernel stats, then the SQ counters (each pass alone)
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/${1:-longprof}
mkdir -p $OUT
ARGS="--reads 2048 --alleles 16 --reps 1"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o v --output-format csv -- python3 scripts/map_long_probe.py $ARGS > $OUT/probe_prof.json 2> $OUT/prof.err; echo "prof rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/prof/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r["Name"][:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d $OUT/pmc_sq -o q --output-format csv -- python3 scripts/map_long_probe.py $ARGS > $OUT/pmc_sq.log 2>&1; echo "sq rc=$?"
timeout 900 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/pmc_sq2 -o q --output-format csv -- python3 scripts/map_long_probe.py $ARGS > $OUT/pmc_sq2.log 2>&1; echo "sq2 rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for d in ("pmc_sq", "pmc_sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(sys.argv[1] + "/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        if "map_long" in k:
            print(k, dict(v))
PY
