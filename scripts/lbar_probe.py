"""How many distinct values does a row of the likelihood matrix take across the alleles? (the threshold-Gram form of
run_filter — sum_r max(M[a][r], M[b][r]) as a weighted Gram matrix of 0/1 indicator columns, one per (read, distinct value) — is a
dense contraction MFMA could run only if that number, L-bar, is small.) CPU oracle on a sample of the synthetic workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from locityper_amd import api, synth
from tests import oracle_ffi as O

def main():
    A, pairs, total = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    L = synth.SynthLocus(A, total)
    p = api.resolve_params(api.default_params(), L.bg)
    ol = O.OracleLocus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    oa = ol.load(L.reads(0, pairs))
    M = oa.best_aln_matrix()                      # [A][n_good]
    d = np.array([len(np.unique(M[:, j])) for j in range(M.shape[1])])
    print(f"A={A} good pairs={M.shape[1]}: distinct values per read: mean {d.mean():.2f} median {np.median(d):.0f} p90 {np.percentile(d, 90):.0f} "
          f"max {d.max()}; indicator columns per read (L-bar - 1) = {d.mean() - 1:.2f}; "
          f"Gram form work ratio vs max-add: {(d.mean() - 1):.1f} x (A^2/2) MACs per read against A^2/2 max-adds")

main()
