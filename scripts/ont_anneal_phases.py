"""Developer measurement (developer build of the library: make -C locityper_amd/csrc DIAG=1): the annealing stage of configs[2] on the
path the reference runs — counted alignments given, resident — with the kernel's shader-clock phases (knob solve_anneal_timing):
   python3 scripts/ont_anneal_phases.py [reads] [--lib path]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locityper_amd import _lib
if "--lib" in sys.argv:
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
else:
    _lib.use_diag_build()
from locityper_amd import api
from bench_legs import long_reads as LR

n = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else 1_000_000
ctx = api.Context(0)
if "--lib" not in sys.argv:
    ctx.set_knob("solve_anneal_timing", 1)
t0 = time.time()
out = LR.ont_whole_path(ctx, n, progress=lambda m: print(f"[{time.time() - t0:6.1f} s] {m}", file=sys.stderr, flush=True), counted=True, chunk=8192)
print(json.dumps({k: out[k] for k in ("reads", "reads_per_s", "seconds", "score_call_s", "solve_call_s", "kernel_ms", "all_calls_equal_truth")}))
