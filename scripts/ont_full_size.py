"""BASELINE.json configs[2] at its stated size — 1 000 000 10-kb ONT reads x 256 alleles — one line of JSON on stdout:
    python3 scripts/ont_full_size.py [reads]                    from the bases alone (the build's own mapper, (f)2), streamed, to the prefilter call
    python3 scripts/ont_full_size.py [reads] --whole-path [--chunk N]   the path the reference runs on GIVEN alignments, records + CIGARs streamed
    python3 scripts/ont_full_size.py [reads] --whole-path --counted   the same with counted alignments, resident (4 GB at full size)
(bench_legs/long_reads.py holds the legs; --check compares eight chains of each solver with the oracle on the scored batch)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locityper_amd import api
from bench_legs import long_reads as LR

chunk = 32768                                       # reads per streamed chunk (--chunk N): 22 GB of records + CIGARs a chunk at 256 alleles
if "--chunk" in sys.argv:
    i = sys.argv.index("--chunk"); chunk = int(sys.argv[i + 1]); del sys.argv[i:i + 2]
args = [a for a in sys.argv[1:] if not a.startswith("--")]
flags = {a for a in sys.argv[1:] if a.startswith("--")}
n = int(args[0]) if args else 1_000_000
ctx = api.Context(0)
t0 = time.time()
say = lambda m: print(f"[{time.time() - t0:7.1f} s] {m}", file=sys.stderr, flush=True)
if "--whole-path" in flags:
    checker = None
    if "--check" in flags:
        from bench_legs.cpu import whole_path_chains_check as checker
    out = LR.ont_whole_path(ctx, n, progress=say, checker=checker, counted="--counted" in flags, chunk=chunk)
else:
    out, _ = LR.ont_from_bases_stream(ctx, n, progress=say)
out["wall_s_incl_generation"] = time.time() - t0
print(json.dumps(out))
