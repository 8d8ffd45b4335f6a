"""Developer check: the alignment of a 10-kb ONT read on an allele as the long route maps it directly against the same (read, allele)
reached by transfer from a basis allele: positions, operation counts, clipped ends. python3 scripts/ont_basis_cigar_diag.py [reads] [basis]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from locityper_amd import api, cdefs, synth, legs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
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
OPS = "MIDNSHP=X"


def summary(words):
    ops = [(int(w) & 15, int(w) >> 4) for w in words]
    cnt = {}
    for o, l in ops: cnt[OPS[o]] = cnt.get(OPS[o], 0) + l
    lead = ops[0][1] if ops and OPS[ops[0][0]] == "S" else 0
    tail = ops[-1][1] if ops and OPS[ops[-1][0]] == "S" else 0
    head = " ".join(f"{l}{OPS[o]}" for o, l in ops[:6]); end = " ".join(f"{l}{OPS[o]}" for o, l in ops[-6:])
    return cnt, lead, tail, head, end


def run(basis, recover):
    api.build_map_index(loc, basis, k=mp.k)
    aa = api.AllAlignments(loc, n, (int(fq.n_bases) + 2048) // 32 * 32, n * len(basis) * 2 + 1024, rb // 3 * len(basis) + 4096)
    api.map_append(aa, fq, mp)
    aa.score()
    if recover: aa.recover()
    out = aa.records()
    M = aa.best_aln_matrix(); st = aa.status()[0]
    aa.close()
    return out, M, st


basis, worst = legs.choose_basis(H, A, nb)
isb = np.zeros(A, bool); isb[basis] = True
(fo, fr, fco, fc), Mf, stf = run(list(range(A)), False)
(bo, br, bco, bc), Mb, stb = run(basis, True)
ixf = np.cumsum(stf == cdefs.READ_GOOD) - 1; ixb = np.cumsum(stb == cdefs.READ_GOOD) - 1
shown = 0
for r in range(n):
    if stf[r] != cdefs.READ_GOOD or stb[r] != cdefs.READ_GOOD: continue
    fa = {int(x["contig"]): i for i, x in enumerate(fr[int(fo[r]):int(fo[r + 1])])}
    ba = {int(x["contig"]): i for i, x in enumerate(br[int(bo[r]):int(bo[r + 1])])}
    for a in sorted(set(fa) & set(ba)):
        if isb[a]: continue
        x = fr[int(fo[r]) + fa[a]]; y = br[int(bo[r]) + ba[a]]
        wx = fc[int(fco[r]) + int(x["cigar_rel"]): int(fco[r]) + int(x["cigar_rel"]) + int(x["n_cigar"])]
        wy = bc[int(bco[r]) + int(y["cigar_rel"]): int(bco[r]) + int(y["cigar_rel"]) + int(y["n_cigar"])]
        cx, lx, tx, hx, ex = summary(wx); cy, ly, ty, hy, ey = summary(wy)
        print(f"read {r} allele {a}: matrix direct {Mf[a, ixf[r]]:.2f} transferred {Mb[a, ixb[r]]:.2f}")
        print(f"   direct      pos {int(x['pos'])} flags {int(x['flags'])} ops {cx} clip {lx}/{tx}   [{hx} ... {ex}]")
        print(f"   transferred pos {int(y['pos'])} flags {int(y['flags'])} ops {cy} clip {ly}/{ty}   [{hy} ... {ey}]")
        shown += 1
        break
    if shown >= 12: break
