"""Developer check: likelihood-matrix entries of 10-kb ONT reads reached by transfer from a basis against the same entries from a direct alignment
(long route onto all alleles): python3 scripts/ont_basis_diag.py [reads] [basis]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from locityper_amd import api, cdefs, synth, legs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
A = 256
ctx = api.Context(0)
L = synth.SynthLocus(A, n, seed=synth.SEED + 77, technology=cdefs.TECH_NANOPORE, read_len=10_000)
p = api.resolve_params(api.default_params(), L.bg)
loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
H = L.hap_alns()
loc.set_hap_alns(H, transfer_fails=100, max_div=0.1)
mp = api.map_params(long_reads=True)
if os.environ.get('LCTY_WFA_SCORES'): mp.match, mp.mismatch, mp.gap_open, mp.gap_extend = 2, 6, 13, 1
fq = synth.sequencer_orientation(L.reads(0, n, primaries_only=True))
rb = int(fq.mate_len.sum())


def run(basis, recover):
    api.build_map_index(loc, basis, k=mp.k)
    aa = api.AllAlignments(loc, n, (int(fq.n_bases) + 2048) // 32 * 32, n * len(basis) * 2 + 1024, rb // 3 * len(basis) + 4096)
    api.map_append(aa, fq, mp)
    aa.score()
    nrec = aa.recover() if recover else 0
    M = aa.best_aln_matrix()            # [A][n_good]
    st = aa.status()[0]
    sc = aa.run_filter()
    aa.close()
    return M, st, sc, nrec


basis, worst = legs.choose_basis(H, A, nb)
print("basis", basis, "largest divergence to the basis %.4f" % worst)
Mf, stf, scf, _ = run(list(range(A)), False)
Mb, stb, scb, nrec = run(basis, True)
gts = api.generate_genotypes(A, 2)
print("truth", L.true_genotype, "full best", gts[int(np.argmax(scf))], "basis best", gts[int(np.argmax(scb))], "recovered", nrec)
good = (stf == cdefs.READ_GOOD) & (stb == cdefs.READ_GOOD)
ixf = np.cumsum(stf == cdefs.READ_GOOD) - 1; ixb = np.cumsum(stb == cdefs.READ_GOOD) - 1
rows = np.nonzero(good)[0]
D = Mb[:, ixb[rows]] - Mf[:, ixf[rows]]          # basis route minus direct, [A][reads]
isb = np.zeros(A, bool); isb[basis] = True
print("reads compared", len(rows))
print("basis alleles: mean diff %.3f, min %.3f, max %.3f" % (D[isb].mean(), D[isb].min(), D[isb].max()))
print("other alleles: mean diff %.3f, p10 %.3f, median %.3f, p90 %.3f, min %.3f, max %.3f" % (D[~isb].mean(), *np.percentile(D[~isb], [10, 50, 90]), D[~isb].min(), D[~isb].max()))
t = L.true_genotype
for a in t:
    print("true allele", a, "in basis" if isb[a] else "not in basis", "mean diff %.3f" % D[a].mean())
