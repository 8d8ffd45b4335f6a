"""BASELINE.json configs[2] at its stated size — 1 M synthetic 10-kb ONT reads x 256 alleles, from the bases alone to the prefilter call — through
the bench's `long_reads_stream` leg (locityper_amd/legs.py::ont_from_bases_stream): python3 scripts/ont_full_size.py [reads] > line.json"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locityper_amd import api, legs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
t0 = time.time()
ctx = api.Context(0)
out, _ = legs.ont_from_bases_stream(ctx, n, progress=lambda m: print(f"[{time.time() - t0:7.1f} s] {m}", file=sys.stderr, flush=True))
out["wall_s_with_generation"] = time.time() - t0
print(json.dumps(out))
