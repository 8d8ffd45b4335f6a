"""Diagnostic: per read, ln-probability of the alignment on the allele the read was drawn from — as the generator's primary, and as
recovered from alignments mapped onto a basis that does not hold that allele."""
import argparse, json, sys
sys.path.insert(0, ".")
import numpy as np
from locityper_amd import api, cdefs, synth

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=256)
ap.add_argument("--alleles", type=int, default=256)
ap.add_argument("--basis", type=int, default=16)
a = ap.parse_args()
ctx = api.Context(0)
A = a.alleles
L = synth.SynthLocus(A, a.reads, seed=synth.SEED + 77, technology=cdefs.TECH_NANOPORE, read_len=10_000)
p = api.resolve_params(api.default_params(), L.bg)
loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
loc.set_hap_alns(L.hap_alns(), transfer_fails=100, max_div=0.1)
prim = L.reads(0, a.reads, primaries_only=True)
origin = [int(prim.recs["contig"][int(prim.aln_off[r])]) for r in range(a.reads)]
ao = api.AllAlignments.load(loc, prim)
ao.recover()
off0, pa0 = ao.pair_alns()
basis = list(range(0, A, max(1, A // a.basis)))[:a.basis]
fq = synth.sequencer_orientation(prim)
mp = api.map_params(long_reads=True)
api.build_map_index(loc, basis, k=mp.k)
rb = int(fq.mate_len.sum())
am = api.AllAlignments(loc, a.reads, (int(fq.n_bases) + 2048) // 32 * 32, a.reads * len(basis) * 2 + 1024, rb // 3 * len(basis) + 4096)
api.map_append(am, fq, mp)
am.score()
offm, pam = am.pair_alns()
am.recover()
off1, pa1 = am.pair_alns()
OP = "MIDNSHP=X"
aln_off, recs, cig_off, cigar = am.records()
shown = 0
for r in range(a.reads):
    o = origin[r]
    if o in basis:
        continue
    e0 = {int(x["contig"]): float(x["ln_prob"]) for x in pa0[int(off0[r]):int(off0[r + 1])]}
    e1 = {int(x["contig"]): float(x["ln_prob"]) for x in pa1[int(off1[r]):int(off1[r + 1])]}
    em = {int(x["contig"]): float(x["ln_prob"]) for x in pam[int(offm[r]):int(offm[r + 1])]}
    if not e1 or not em:
        continue
    best_b = max(em, key=em.get)
    row = {"read": r, "origin": o, "gen_primary_lnp": e0.get(o), "recovered_origin_lnp": e1.get(o), "best_basis": best_b, "best_basis_lnp": em[best_b],
           "n_mapped": len(em), "n_after": len(e1)}
    # the records of this read after recovery: edit operations on origin / best basis
    for i in range(int(aln_off[r]), int(aln_off[r + 1])):
        c = int(recs["contig"][i])
        if c in (o, best_b):
            w = cigar[int(cig_off[r]) + int(recs["cigar_rel"][i]):][:int(recs["n_cigar"][i])]
            ops = {k: int(sum(int(x) >> 4 for x in w if OP[int(x) & 15] == k)) for k in "=XIDS"}
            row["cigar_origin" if c == o else "cigar_best_basis"] = ops
    print(json.dumps(row))
    shown += 1
    if shown >= 8:
        break
