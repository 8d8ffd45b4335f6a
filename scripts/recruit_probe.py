"""Developer probe for minimizer read recruitment: whole-genome-like input, most read pairs foreign.
    python scripts/recruit_probe.py [n_pairs] [n_loci] [fraction_from_loci]
Prints one JSON line: read pairs/s of recruit_kernel (inputs resident) and of the whole call (the oracle's rate is in bench.py's cpu_baseline)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from locityper_amd import api, cdefs, synth
from locityper_amd.cdefs import ReadsChunk

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 8
FRAC = float(sys.argv[3]) if len(sys.argv) > 3 else 0.02

ctx = api.Context(0)
prm = api.recruit_params()
T = api.Targets(ctx, prm)
loci = []
t0 = time.time()
for l in range(NL):
    L = synth.SynthLocus(64, 1000, seed=synth.SEED + 100 + l, base_len=50_000)
    T.add_locus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k)
    loci.append(L)
n_minim = T.finalize()
t_build = time.time() - t0

# reads: FRAC of the pairs come from the loci (error-free 150 + 150 bases, mate 2 reverse-complemented as a sequencer reports it),
# the rest is random sequence
rng = np.random.default_rng(1)
n_loc = int(N * FRAC)
per = max(1, n_loc // NL)
CODE = np.full(256, 0, dtype=np.uint8); CODE[ord("C")] = 1; CODE[ord("G")] = 2; CODE[ord("T")] = 3


def pack(codes):                                             # [n, 150] base codes -> [n, 10] words (160-base slots)
    pad = np.zeros((codes.shape[0], 160), dtype=np.uint32); pad[:, :150] = codes
    return (pad.reshape(-1, 10, 16) << (2 * np.arange(16, dtype=np.uint32))).sum(axis=2).astype(np.uint32)


chunks = []
for l in range(NL):
    L = loci[l]
    al = rng.integers(0, 64, per)
    st = np.array([rng.integers(0, int(L.seq_off[a + 1] - L.seq_off[a]) - 600) for a in al], dtype=np.int64) + L.seq_off[al].astype(np.int64)
    idx = st[:, None] + np.arange(150)[None, :]
    m1 = CODE[L.seqs[idx]]
    m2 = 3 - CODE[L.seqs[idx + 350]][:, ::-1]
    words = np.stack([pack(m1), pack(m2)], axis=1).reshape(-1)
    chunks.append(ReadsChunk(np.full(2 * per, 150, dtype=np.uint32), np.arange(2 * per + 1, dtype=np.uint64) * 160, words,
                             np.zeros(per * 2 * 5, dtype=np.uint32), np.zeros(per + 1, dtype=np.uint64), np.zeros(0, dtype=cdefs.ALN_REC_DTYPE),
                             np.zeros(per + 1, dtype=np.uint64), np.zeros(0, dtype=np.uint32)))
n_loc = per * NL
n_rand = N - n_loc
# packed random pairs: 150 + 150 bases, 160-base slots
words = rng.integers(0, 1 << 32, size=n_rand * 2 * 10, dtype=np.uint64).astype(np.uint32)
mate_len = np.full(2 * n_rand, 150, dtype=np.uint32)
mate_off = (np.arange(2 * n_rand + 1, dtype=np.uint64) * 160)
rand = ReadsChunk(mate_len, mate_off, words, np.zeros(n_rand * 2 * 5, dtype=np.uint32), np.zeros(n_rand + 1, dtype=np.uint64),
                  np.zeros(0, dtype=cdefs.ALN_REC_DTYPE), np.zeros(n_rand + 1, dtype=np.uint64), np.zeros(0, dtype=np.uint32))
res = {"n_pairs": N, "n_loci": NL, "alleles_per_locus": 64, "minimizers": n_minim, "targets_build_s": round(t_build, 2)}
ctx.timing_reset()
t0 = time.time()
cnt_r, _ = T.recruit(rand, paired=True)
t_rand = time.time() - t0
rec_loc = 0
for l, ch in enumerate(chunks):
    c, out = T.recruit(ch, paired=True)
    rec_loc += int(np.sum([(l in out[i, :c[i]]) for i in range(len(c))]))
n_launch, ms = ctx.timing(api.K_RECRUIT)
res.update({"kernel_ms": round(ms, 3), "launches": n_launch, "pairs_per_s_kernel": round(N / (ms * 1e-3)),
            "packed_input_GBs": round(N * 80 / (ms * 1e-3) / 1e9, 1), "pairs_per_s_call_random_part": round(n_rand / t_rand),
            "foreign_pairs_recruited": int(np.count_nonzero(cnt_r)), "locus_pairs": n_loc, "locus_pairs_recruited_to_their_locus": rec_loc})
print(json.dumps(res))
