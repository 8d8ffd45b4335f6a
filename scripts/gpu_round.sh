#!/bin/bash
# One gpurun call of a round, parameterised (replaces the one-off launchers of round 2):
#   scripts/gpu_round.sh <tag> [tests] [bench [bench args...]] 
# steps are chosen by words in $2: t = pytest -m gpu, b = plain bench.py line, p = rocprofv3 --kernel-trace --stats of a short bench,
# f / w = the FETCH_SIZE / WRITE_SIZE passes (each alone: never combined with other traces), q = SQ counters pass.
# Everything lands under gpurun_out/<tag>/; copy what is to be judged into profiles/.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
TAG=${1:-round}; STEPS=${2:-tb}; shift; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
python3 scripts/sources_sha.py > "$OUT/sources.sha16"        # which kernels these passes saw (bench.py compares)
LEAN="--cpu-sample 0 --exact-sample 0 --ont-whole-path-sample 0 --ont-stream-sample 0 --recovery-sample 0 --recruit-sample 0 --map-sample 0 --many-alleles-sample 0 --ont-sample 0 --ont-map-sample 0"
case "$STEPS" in *t*) timeout 1500 python3 -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" "$OUT/pytest.log" | tail -15;; esac
case "$STEPS" in *b*) timeout 1200 python3 bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; cut -c1-600 "$OUT/bench.json";; esac
case "$STEPS" in *l*) timeout 900 python3 bench.py $LEAN "$@" > "$OUT/bench_lean.json" 2> "$OUT/bench_lean.err"; echo "lean bench rc=$?"; cut -c1-600 "$OUT/bench_lean.json";; esac
case "$STEPS" in *p*) timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o v --output-format csv -- python3 bench.py --steps 4 --warmup 2 $LEAN "$@" > "$OUT/bench_under_rocprof.json" 2> "$OUT/prof.err"; echo "prof rc=$?";
   python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
;; esac
case "$STEPS" in *f*) timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o f --output-format csv -- python3 bench.py --steps 1 --warmup 0 $LEAN "$@" > "$OUT/pmc_fetch.log" 2>&1; echo "fetch rc=$?";; esac
case "$STEPS" in *w*) timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o w --output-format csv -- python3 bench.py --steps 1 --warmup 0 $LEAN "$@" > "$OUT/pmc_write.log" 2>&1; echo "write rc=$?";; esac
case "$STEPS" in *q*) timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace -d "$OUT/pmc_sq" -o q --output-format csv -- python3 bench.py --steps 1 --warmup 0 $LEAN "$@" > "$OUT/pmc_sq.log" 2>&1; echo "sq rc=$?";; esac
ls "$OUT"
