"""Developer measurement: lcty_solve_given (the per-call `Solver::solve` route of the Rust shim) at the size of BASELINE configs[1]:
   python3 scripts/solve_given_probe.py [pairs] [alleles]
One GenotypeAlignments of the true genotype over all read pairs (built by the oracle from the device's products, tweaked with a key of
its own), handed over as arrays; per solver: the call's wall time (uploads included), the same with the caller's own depth rows
(lcty_solve_given_tables), four calls at once from four threads, and the likelihood against the oracle's chain on that object."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from locityper_amd import api, synth, cdefs
from tests import oracle_ffi as O


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    A = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    L = synth.SynthLocus(A, pairs, seed=synth.SEED)
    p = api.resolve_params(api.default_params(), L.bg)
    ctx = api.Context(0)
    loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    aa = None
    for lo in range(0, pairs, 32768):
        ch = L.reads(lo, min(32768, pairs - lo))
        if aa is None:
            f = 1.05 * pairs / ch.n_pairs
            aa = api.AllAlignments(loc, pairs, (int(ch.n_bases * f) + 2048) // 32 * 32, int(len(ch.recs) * f) + 4096, 0)
        aa.append(ch, counted=True)
    aa.score()
    st, w, unm, _ = aa.status()
    off, pa = aa.pair_alns()
    ol = O.OracleLocus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    ol.inject_tables(loc.depth_lut(), loc.window_weights())
    ol.inject_depth_table(loc.depth_table(8192))
    oa = O.alns_from_arrays(A, st, w, unm, off, pa)
    del pa, off
    t0 = time.perf_counter()
    g = O.OracleGtAlns(ol, oa, tuple(int(x) for x in L.true_genotype))
    g.apply_tweak(12345)
    a = g.arrays()
    gc, ww = g.window_distr()
    v = dict(read_ixs=a["read_ixs"], ln_prob=a["ln_prob"], windows=a["windows"], window_gc=gc, window_weight=ww,
             depth_contrib=1.0 + p.lik_skew, aln_contrib=1.0 - p.lik_skew)
    print(f"GenotypeAlignments of {g.n_reads} reads, {g.n_alns} locations, {g.n_nontrivial} non-trivial, {g.n_windows} windows "
          f"(oracle: {time.perf_counter() - t0:.2f} s)", flush=True)
    deepest = api.solve_given(ctx, solver=None, rng_state=None, deepest_only=True, **v)
    rows = np.ascontiguousarray(loc.depth_table(deepest + 1)[:, :deepest + 1])
    for kind, name in ((cdefs.SOLVER_GREEDY, "greedy"), (cdefs.SOLVER_ANNEAL, "annealing")):
        sv = api.default_solver(kind)
        api.solve_given(loc, solver=sv, rng_state=api.rng_seed_from_u64(1), **v)                    # first call: the slot's buffers
        t0 = time.perf_counter()
        state = api.rng_seed_from_u64(7)
        seed = api.rng_next_u64(state.copy())
        lik, assgn, parts = api.solve_given(loc, solver=sv, rng_state=state, **v)
        t_call = time.perf_counter() - t0
        t0 = time.perf_counter()
        lik_t, assgn_t, _ = api.solve_given(ctx, solver=sv, rng_state=api.rng_seed_from_u64(7), tables=rows, tables_id=1, **v)
        t_tab = time.perf_counter() - t0
        out = [None] * 4
        def work(i):
            out[i] = api.solve_given(loc, solver=sv, rng_state=api.rng_seed_from_u64(100 + i), **v)[0]
        ths = [threading.Thread(target=work, args=(i,)) for i in range(4)]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        t_four = time.perf_counter() - t0
        t0 = time.perf_counter()
        olik, oassgn, _ = g.solve(sv, seed)
        t_orc = time.perf_counter() - t0
        print(f"{name}: lcty_solve_given {1e3 * t_call:.1f} ms per call ({1e3 * t_tab:.1f} ms with the caller's depth rows; four calls at once from "
              f"four threads {1e3 * t_four:.1f} ms); likelihood {lik:.6f}, the oracle's chain on the same object {olik:.6f} "
              f"({t_orc:.2f} s on one core), same assignment: {bool(np.array_equal(assgn, oassgn))}, tables form equal: "
              f"{bool(lik == lik_t and np.array_equal(assgn, assgn_t))}", flush=True)


main()
