"""Developer tool: time the scoring / prefilter kernels (optionally under LCTY_DBG ablations)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from locityper_amd import api, synth
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
A = int(sys.argv[2]) if len(sys.argv) > 2 else 256
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["0"]
L = synth.SynthLocus(A, 1_000_000)
p = api.resolve_params(api.default_params(), L.bg)
ctx = api.Context(0)
loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
chunks = [L.reads(lo, min(32768, pairs - lo)) for lo in range(0, pairs, 32768)]
aa = api.AllAlignments(loc, pairs, sum(c.n_bases for c in chunks), sum(len(c.recs) for c in chunks), sum(len(c.cigar) for c in chunks))
for c in chunks: aa.append(c)
for m in modes:
    os.environ["LCTY_DBG"] = m
    aa.score(); ctx.synchronize(); ctx.timing_reset()
    for _ in range(3): aa.score()
    n, ms = ctx.timing(api.K_SCORE)
    line = f"dbg={m} score {ms/n:.3f} ms ({pairs/(ms/n)*1e3/1e6:.2f} Mpairs/s)"
    if m == "0":
        aa.prefilter_async(); ctx.synchronize(); ctx.timing_reset()
        for _ in range(3): aa.prefilter_async()
        n2, ms2 = ctx.timing(api.K_PREFILTER)
        line += f"  prefilter {ms2/n2:.3f} ms"
    print(line, flush=True)
