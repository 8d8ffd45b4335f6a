"""Developer measurement: is solve_init_kernel bound by its table reads? 5 000 chains of DIFFERENT genotypes against 5 000 chains of ONE
genotype (every workgroup streams the same two table rows: L2 / MALL hits) and of genotypes sharing their first allele."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from locityper_amd import api, synth, cdefs

def main():
    A, pairs, n = 256, 1_000_000, 5000
    L = synth.SynthLocus(A, pairs, seed=synth.SEED)
    p = api.resolve_params(api.default_params(), L.bg)
    ctx = api.Context(0)
    loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    aa = None
    for lo in range(0, pairs, 32768):
        ch = L.reads(lo, min(32768, pairs - lo))
        if aa is None:
            f = 1.05 * pairs / ch.n_pairs
            aa = api.AllAlignments(loc, pairs, (int(ch.n_bases * f) + 2048) // 32 * 32, int(len(ch.recs) * f) + 4096, int(len(ch.cigar) * f) + 65536)
        aa.append(ch)
    aa.score()
    sc = aa.run_filter()
    gts = api.generate_genotypes(A, 2)
    order = np.argsort(-sc, kind="stable")
    top = gts[order[:n]]
    one = np.repeat(top[:1], n, axis=0)
    a0 = int(top[0][0])
    shared = np.array([[min(a0, b), max(a0, b)] for b in (np.arange(n) % A)], dtype=np.uint16)
    sv = api.default_solver(cdefs.SOLVER_GREEDY)
    sv.plato_size = 1                                                        # the loop ends at once: the stage is its initialisation
    seeds = api.chain_seeds(7, n)
    api.solve_stage(aa, top, sv, 1, seeds)
    for name, g in (("5000 different genotypes", top), ("one genotype 5000 times", one), ("5000 genotypes sharing an allele", shared), ("5000 different genotypes", top)):
        ctx.timing_reset()
        api.solve_stage(aa, np.ascontiguousarray(g), sv, 1, seeds)
        print(f"{name}: init {ctx.timing(api.K_SOLVE_INIT)[1]:.1f} ms, loop {ctx.timing(api.K_SOLVE)[1]:.1f} ms", flush=True)

main()
