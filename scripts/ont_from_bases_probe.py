"""configs[2] from bases alone vs from the generator's primary records: where does the genotype the reads were drawn from rank after
mapping onto a basis + alignment recovery + prefilter?  usage: python3 scripts/ont_from_bases_probe.py [--reads N] [--alleles A] [--basis B]"""
import argparse, json, sys
sys.path.insert(0, ".")
import numpy as np
from locityper_amd import api, cdefs, synth

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=1024)
ap.add_argument("--alleles", type=int, default=256)
ap.add_argument("--basis", type=int, default=16)
ap.add_argument("--seed", type=int, default=synth.SEED + 77)
a = ap.parse_args()
ctx = api.Context(0)
A = a.alleles
L = synth.SynthLocus(A, a.reads, seed=a.seed, technology=cdefs.TECH_NANOPORE, read_len=10_000)
p = api.resolve_params(api.default_params(), L.bg)
loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
loc.set_hap_alns(L.hap_alns(), transfer_fails=100, max_div=0.1)
gts = api.generate_genotypes(A, 2)
truth_ix = [i for i, g in enumerate(gts) if tuple(int(x) for x in g) == tuple(L.true_genotype)][0]
chunk = 256
prim = [L.reads(lo, min(chunk, a.reads - lo), primaries_only=True) for lo in range(0, a.reads, chunk)]


def rank_of_truth(aa, label):
    sc = aa.run_filter()
    order = np.argsort(-sc)
    rank = int(np.where(order == truth_ix)[0][0])
    print(json.dumps({"what": label, "good": aa.n_good(), "rank_of_truth": rank, "best": [int(x) for x in gts[int(order[0])]], "truth": [int(x) for x in L.true_genotype],
                      "score_best": float(sc[order[0]]), "score_truth": float(sc[truth_ix])}))


ao = api.AllAlignments.load(loc, prim)
ao.recover()
rank_of_truth(ao, "generator's primaries + recovery")
ao.close()
basis = list(range(0, A, max(1, A // a.basis)))[:a.basis]
tb = set(int(x) for x in L.true_genotype)
for label, bs in (("mapped onto the basis + recovery", basis), ("mapped onto the basis + the two true alleles + recovery", sorted(set(basis) | tb))):
    fq = [synth.sequencer_orientation(c) for c in prim]
    mp = api.map_params(long_reads=True)
    api.build_map_index(loc, bs, k=mp.k)
    nb = sum(int(c.n_bases) for c in fq); rb = sum(int(c.mate_len.sum()) for c in fq)
    am = api.AllAlignments(loc, a.reads, (nb + 2048) // 32 * 32, a.reads * len(bs) * 2 + 1024, rb // 3 * len(bs) + 4096)
    for c in fq:
        api.map_append(am, c, mp)
    am.score()
    before = int(am.pair_alns()[0][-1])
    rec = am.recover()
    print(json.dumps({"basis": len(bs), "mapped_pair_alns": before, "recovered": int(rec)}))
    rank_of_truth(am, label)
    am.close()
