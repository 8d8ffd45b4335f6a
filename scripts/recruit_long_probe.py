"""Developer probe: recruitment of long single reads (recruit_long_read, one wavefront per read).
    python scripts/recruit_long_probe.py [n_reads] [read_len]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from locityper_amd import api, cdefs, synth
from locityper_amd.cdefs import ReadsChunk

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
LEN = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
ctx = api.Context(0)
prm = api.recruit_params(technology=cdefs.TECH_NANOPORE, paired=False)
T = api.Targets(ctx, prm)
L = synth.SynthLocus(64, 1000, seed=synth.SEED + 100, base_len=50_000)
T.add_locus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k)
n_minim = T.finalize()
slot = (LEN + 31) // 32 * 32
rng = np.random.default_rng(2)
words = rng.integers(0, 1 << 32, size=N * slot // 16, dtype=np.uint64).astype(np.uint32)
# a tenth of the reads are copies of allele stretches
CODE = np.zeros(256, dtype=np.uint8); CODE[ord("C")] = 1; CODE[ord("G")] = 2; CODE[ord("T")] = 3
n_loc = N // 10
al = rng.integers(0, 64, n_loc)
for i in range(n_loc):
    a = int(al[i]); st = int(L.seq_off[a]) + int(rng.integers(0, int(L.seq_off[a + 1] - L.seq_off[a]) - LEN))
    codes = np.zeros(slot, dtype=np.uint32); codes[:LEN] = CODE[L.seqs[st:st + LEN]]
    words[i * slot // 16:(i + 1) * slot // 16] = (codes.reshape(-1, 16) << (2 * np.arange(16, dtype=np.uint32))).sum(axis=1).astype(np.uint32)
mate_len = np.zeros(2 * N, dtype=np.uint32); mate_len[0::2] = LEN
mate_off = np.zeros(2 * N + 1, dtype=np.uint64); mate_off[1::2] = np.arange(1, N + 1, dtype=np.uint64) * slot; mate_off[2::2] = mate_off[1::2][:N]
mate_off[0] = 0
mate_off[1:] = np.repeat(np.arange(1, N + 1, dtype=np.uint64) * slot, 2)[:2 * N]
ch = ReadsChunk(mate_len, mate_off, words, np.zeros(N * slot // 32, dtype=np.uint32), np.zeros(N + 1, dtype=np.uint64),
                np.zeros(0, dtype=cdefs.ALN_REC_DTYPE), np.zeros(N + 1, dtype=np.uint64), np.zeros(0, dtype=np.uint32))
ctx.timing_reset()
t0 = time.time()
cnt, out = T.recruit(ch, paired=False)
t_call = time.time() - t0
_, ms = ctx.timing(api.K_RECRUIT)
print(json.dumps({"n_reads": N, "read_len": LEN, "minimizers": n_minim, "kernel_ms": round(ms, 3), "reads_per_s_kernel": round(N / (ms * 1e-3)),
                  "Gbases_per_s_kernel": round(N * LEN / (ms * 1e-3) / 1e9, 2), "call_s": round(t_call, 3),
                  "locus_reads": n_loc, "locus_reads_recruited": int(np.count_nonzero(cnt[:n_loc])), "foreign_recruited": int(np.count_nonzero(cnt[n_loc:]))}))
