"""Time the long route of candidate generation (lcty_map_reads on 10-kb ONT reads): reads x basis alleles alignments per second.
usage: python3 scripts/map_long_probe.py [--reads N] [--alleles A] [--basis B] [--read-len L] [--stride S] [--reps R]"""
import argparse, json, os, sys, time
sys.path.insert(0, ".")
if "--trace" in sys.argv:                                # the mapper's trace exists in the developer build only (make DIAG=1)
    from locityper_amd import _lib
    _lib.use_diag_build()
if "--lib" in sys.argv:                                  # a library built by hand
    from locityper_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1]); del sys.argv[sys.argv.index("--lib"):sys.argv.index("--lib") + 2]
import numpy as np
from locityper_amd import api, cdefs, synth

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=1024)
ap.add_argument("--alleles", type=int, default=16)
ap.add_argument("--basis", type=int, default=0)
ap.add_argument("--read-len", type=int, default=10_000)
ap.add_argument("--base-len", type=int, default=40_000)
ap.add_argument("--stride", type=int, default=16)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--trace", action="store_true")
ap.add_argument("--chain-back", type=int, default=0)
a = ap.parse_args()
ctx = api.Context(0)
if a.trace:
    ctx.set_knob("map_trace", 1)
L = synth.SynthLocus(a.alleles, a.reads, seed=synth.SEED + 5, technology=cdefs.TECH_NANOPORE, read_len=a.read_len, base_len=a.base_len)
p = api.resolve_params(api.default_params(), L.bg)
loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
fq = synth.sequencer_orientation(L.reads(0, a.reads, primaries_only=True))
mp = api.map_params(long_reads=True, stride=a.stride)
if a.chain_back:
    mp.chain_back = a.chain_back
basis = list(range(a.basis or a.alleles))
t0 = time.time(); api.build_map_index(loc, basis, k=mp.k); t_index = time.time() - t0
best = None
bases = int(fq.mate_len.sum())
for rep in range(a.reps):
    aa = api.AllAlignments(loc, a.reads, (int(fq.n_bases) + 2048) // 32 * 32, a.reads * len(basis) * 2 + 1024, bases // 3 * len(basis) + 4096)
    ctx.timing_reset()
    t0 = time.time()
    api.map_append(aa, fq, mp)
    dt = time.time() - t0
    k = ctx.timing(api.K_MAP)
    best = dt if best is None else min(best, dt)
    aa.score()
    n_rec = int(aa.pair_alns()[0][-1])
    aa.close()
print(json.dumps({"reads": a.reads, "basis": len(basis), "read_len": a.read_len, "bases": bases, "index_s": round(t_index, 3), "call_s": round(best, 4),
                  "kernel": k, "pair_alns": n_rec, "alignments_per_s_kernel": round(a.reads * len(basis) / (k[1] * 1e-3), 1),
                  "aligned_bases_per_s_kernel": round(len(basis) * bases / (k[1] * 1e-3), 1), "gcups": round(len(basis) * bases * 33 / (k[1] * 1e-3) / 1e9, 2)}))
