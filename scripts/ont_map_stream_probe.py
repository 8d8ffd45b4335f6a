"""configs[2] from bases alone through a STREAMING batch: chunks of 10-kb ONT reads as sequenced are mapped onto every allele (long route),
appended device to device and scored chunk by chunk (the records of a chunk are dropped once it is scored), then the prefilter calls.
usage: python3 scripts/ont_map_stream_probe.py [--reads N] [--alleles A] [--chunk C]"""
import argparse, json, sys, time
sys.path.insert(0, ".")
import numpy as np
from locityper_amd import api, cdefs, synth

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=8192)
ap.add_argument("--alleles", type=int, default=256)
ap.add_argument("--chunk", type=int, default=1024)
a = ap.parse_args()
ctx = api.Context(0)
A = a.alleles
L = synth.SynthLocus(A, a.reads, seed=synth.SEED + 77, technology=cdefs.TECH_NANOPORE, read_len=10_000)
p = api.resolve_params(api.default_params(), L.bg)
loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
mp = api.map_params(long_reads=True)
t0 = time.time(); api.build_map_index(loc, list(range(A)), k=mp.k); t_index = time.time() - t0
chunks = [synth.sequencer_orientation(L.reads(lo, min(a.chunk, a.reads - lo), primaries_only=True)) for lo in range(0, a.reads, a.chunk)]
cb = max(int(c.n_bases) for c in chunks); rb = max(int(c.mate_len.sum()) for c in chunks)
aa = api.AllAlignments(loc, a.reads, (cb + 2048) // 32 * 32, a.chunk * A * 2 + 1024, rb // 3 * A + 4096, streaming_chunk_pairs=a.chunk)
ctx.timing_reset()
t0 = time.time()
for c in chunks:
    api.map_append(aa, c, mp)
    aa.score()
dt = time.time() - t0
_, ms_map = ctx.timing(api.K_MAP); _, ms_score = ctx.timing(api.K_SCORE)
sc = aa.run_filter()
gts = api.generate_genotypes(A, 2)
best = tuple(int(x) for x in gts[int(np.argmax(sc))])
truth_ix = [i for i, g in enumerate(gts) if tuple(int(x) for x in g) == tuple(L.true_genotype)][0]
print(json.dumps({"reads": a.reads, "alleles": A, "chunk": a.chunk, "index_s": round(t_index, 2), "map_and_score_s": round(dt, 3), "reads_per_s": round(a.reads / dt, 1),
                  "alignments_per_s": round(a.reads * A / dt, 1), "map_kernels_ms": round(ms_map, 1), "score_kernels_ms": round(ms_score, 1), "good_reads": aa.n_good(),
                  "best": best, "truth": [int(x) for x in L.true_genotype], "truth_scores_as_the_best": bool(sc[truth_ix] >= sc.max() - 1e-9 * abs(sc.max()))}))
