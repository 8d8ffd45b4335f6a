"""Developer check: GPU solver stages vs oracle fed with the GPU's own tables / alignments (exact chain parity)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from locityper_amd import api, synth, cdefs
from tests import oracle_ffi as O

def run(n_alleles, n_pairs, base_len, n_gt, attempts):
    L = synth.SynthLocus(n_alleles, n_pairs, seed=31, base_len=base_len)
    p = api.resolve_params(api.default_params(), L.bg)
    ctx = api.Context(0)
    loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    ch = L.reads(0, n_pairs)
    aa = api.AllAlignments.load(loc, ch)
    st, w, unm, uk = aa.status()
    off, pa = aa.pair_alns()
    ol = O.OracleLocus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    ol.inject_tables(loc.depth_lut(), loc.window_weights())
    oa = O.alns_from_arrays(n_alleles, st, w, unm, off, pa)
    gts = api.generate_genotypes(n_alleles, 2)
    sc = aa.run_filter()
    order = np.argsort(-sc)[:n_gt]
    sub = gts[order]
    seeds = api.chain_seeds(2024, len(sub) * attempts)
    for kind in (cdefs.SOLVER_GREEDY, cdefs.SOLVER_ANNEAL):
        solver = api.default_solver(kind)
        t = time.time(); gm, gv, gl = api.solve_stage(aa, sub, solver, attempts, seeds); tg = time.time() - t
        t = time.time(); om, ov, olk = O.solve_stage(ol, oa, sub, solver, attempts, seeds); to = time.time() - t
        d = np.abs(gl - olk)
        print(f"kind={kind} gpu {tg:.3f}s oracle {to:.3f}s max|dlik|={d.max():.3e} rel={(d/np.abs(olk)).max():.3e} "
              f"exact={np.mean(d < 1e-7*np.abs(olk)):.3f} best gpu {sub[np.argmax(gm)]} oracle {sub[np.argmax(om)]} true {L.true_genotype}")
        bad = np.argwhere(d > 1e-6 * np.abs(olk))
        if len(bad): print("  mismatching chains:", bad[:5].tolist(), gl[tuple(bad[0])], olk[tuple(bad[0])])

if __name__ == "__main__":
    run(8, 3000, 20000, 12, 2)
    run(16, 20000, 30000, 40, 3)
