"""Developer probe for alignment recovery (K6): the mapper reports only the primary alignment of each read end, the alignments
to the other alleles are transferred through the haplotype-to-haplotype alignments (transfer.rs:70-140).
    python scripts/recovery_probe.py [n_pairs] [n_alleles] [transfer_fails] [ont]
Prints one JSON line: transfers/s of the transfer kernel, the second scoring pass, agreement with the all-alignments load."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from locityper_amd import api, cdefs, synth

R = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
A = int(sys.argv[2]) if len(sys.argv) > 2 else 256
TF = int(sys.argv[3]) if len(sys.argv) > 3 else 100      # the reference's default (transfer.rs / genotype.rs --transfer-fails)

ctx = api.Context(0)
t0 = time.time()
ONT = len(sys.argv) > 4 and sys.argv[4] == "ont"          # single-end long reads (10 kb, CIGARs of hundreds of operations)
L = synth.SynthLocus(A, R, technology=cdefs.TECH_NANOPORE, read_len=10_000) if ONT else synth.SynthLocus(A, R)
p = api.resolve_params(api.default_params(), L.bg)
loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
H = L.hap_alns()
t1 = time.time()
loc.set_hap_alns(H, transfer_fails=TF, max_div=0.05)
t_set = time.time() - t1
full = L.reads(0, R)
prim = full.primaries()
t_gen = time.time() - t0

aa = api.AllAlignments.load(loc, prim)
ctx.timing_reset()
t1 = time.time()
n_new = aa.recover()                                       # transfer + merge of the record tables + second scoring pass
t_rec = time.time() - t1
n_launch, ms_transfer = ctx.timing(5)                      # LCTY_K_TRANSFER
_, ms_score2 = ctx.timing(0)                               # LCTY_K_SCORE
m_rec = aa.best_aln_matrix()

ab = api.AllAlignments.load(loc, full)
m_full = ab.best_aln_matrix()
same_shape = m_rec.shape == m_full.shape
close = float(np.mean(np.abs(m_rec - m_full) < 1e-6)) if same_shape else None
print(json.dumps({
    "n_pairs": R, "n_alleles": A, "transfer_fails": TF, "hap_alns": len(H), "hap_cigar_words": int(sum(len(e[2]) for e in H)),
    "set_hap_alns_s": round(t_set, 3), "records_in": int(len(prim.recs)), "records_recovered": int(n_new),
    "transfer_kernel_ms": round(ms_transfer, 3), "transfer_launches": int(n_launch), "recover_call_s": round(t_rec, 3),
    "transfers_per_s": round(n_new / (ms_transfer * 1e-3)) if ms_transfer else None,
    "second_score_kernel_ms": round(ms_score2, 3), "n_good": int(m_rec.shape[1]), "n_good_full_table": int(m_full.shape[1]),
    "matrix_cells_equal_to_full_table": close, "generate_s": round(t_gen, 1)}))
