"""The exact solver (the place of the reference's HiGHS / Gurobi back ends, src/solvers/highs.rs:38-134) through the C ABI
against the oracle's exhaustive enumeration of the same model, on instances small enough to enumerate; on larger ones against the
chains (an optimum is never below what greedy or annealing find); the node limit; the assignment it leaves behind."""
import numpy as np
import pytest

from locityper_amd import _lib, api, cdefs, synth
from tests import oracle_ffi as O
from tests.test_gpu_solve import setup

pytestmark = pytest.mark.gpu


def proof():
    """the exact solver asked for a proof of optimality (relative gap 0) instead of the default gap of 1e-4"""
    s = api.default_solver(cdefs.SOLVER_EXACT)
    assert s.init_prob == 1e-4                   # HiGHS' default mip_rel_gap, which the reference leaves alone (highs.rs:103-110)
    s.init_prob = 0.0
    return s


def test_exact_equals_enumeration_on_small_models(gpu_ctx):
    exact = proof()
    assert exact.kind == cdefs.SOLVER_EXACT and exact.node_limit == 20_000_000
    o_exact = O.default_solver(cdefs.SOLVER_EXACT)
    n_cases = 0
    for n_alleles, n_pairs, base_len, seed in ((4, 11, 4000, 9), (5, 13, 3000, 21), (3, 12, 2500, 4), (6, 9, 5000, 33)):
        L, p, loc, aa, ol, oa = setup(gpu_ctx, n_alleles, n_pairs, base_len, seed=seed)
        gts = api.generate_genotypes(n_alleles, 2)
        seeds = api.chain_seeds(100 + seed, 2 * len(gts))
        gm, gv, gl = api.solve_stage(aa, gts, exact, 2, seeds)
        om, ov, olk = O.solve_stage(ol, oa, gts, o_exact, 2, seeds)
        assert np.all(np.isfinite(olk)), "the oracle's enumeration must cover these models"
        assert np.abs(gl - olk).max() <= 1e-9 * np.abs(olk).max()
        assert np.allclose(gm, om, rtol=1e-9)
        # never below the chains of the same attempt (same tweak)
        for kind in (cdefs.SOLVER_GREEDY, cdefs.SOLVER_ANNEAL):
            _, _, cl = api.solve_stage(aa, gts, api.default_solver(kind), 2, seeds)
            assert np.all(cl <= gl + 1e-9 * np.abs(gl))
        n_cases += gl.size
    assert n_cases >= 90
    # other ploidies
    L, p, loc, aa, ol, oa = setup(gpu_ctx, 4, 10, 3000, seed=5)
    for ploidy in (1, 3):
        g = api.generate_genotypes(4, ploidy)[:4]
        s = api.chain_seeds(3, len(g))
        gl = api.solve_stage(aa, g, exact, 1, s)[2]
        olk = O.solve_stage(ol, oa, g, o_exact, 1, s)[2]
        assert np.all(np.isfinite(olk)) and np.abs(gl - olk).max() <= 1e-9 * np.abs(olk).max()


def test_exact_on_larger_models_node_limit_and_assignment(gpu_ctx):
    L, p, loc, aa, ol, oa = setup(gpu_ctx, 6, 400, 6000, seed=12)
    gts = api.generate_genotypes(6, 2)[:6]
    seeds = api.chain_seeds(8, len(gts))
    exact = proof()
    try:
        gl = api.solve_stage(aa, gts, exact, 1, seeds)[2]
    except _lib.LocityperError as e:
        assert e.code == cdefs.ERR_SOLVER and "non-optimal status" in str(e)          # as HiGHS' non-optimal status (highs.rs:113-116)
        gl = None
    if gl is not None:
        for kind in (cdefs.SOLVER_GREEDY, cdefs.SOLVER_ANNEAL):
            cl = api.solve_stage(aa, gts, api.default_solver(kind), 1, seeds)[2]
            assert np.all(cl <= gl + 1e-9 * np.abs(gl))
        # priors shift the likelihood as for every solver (solve.rs:827)
        pri = -np.arange(len(gts), dtype=np.float64)
        assert np.allclose(api.solve_stage(aa, gts, exact, 1, seeds, pri)[2], gl + pri[:, None], rtol=1e-13)
    # with a relative gap (HiGHS' mip_rel_gap; lcty_solver.init_prob of this kind) the search stops as soon as nothing left can beat the
    # incumbent by more than the gap: an answer where the proof of optimality runs out of nodes, never above the optimum, within the gap of it
    loose = api.default_solver(cdefs.SOLVER_EXACT)
    loose.init_prob = 0.25
    ll = api.solve_stage(aa, gts, loose, 1, seeds)[2]
    assert np.all(np.isfinite(ll))
    if gl is not None:
        assert np.all(ll <= gl + 1e-9 * np.abs(gl)) and np.all(ll >= gl - 0.25 * np.abs(gl))
    # a node limit too small for a proof: Error::Solver, never an unproven answer
    tight = proof()
    tight.node_limit = 3
    with pytest.raises(_lib.LocityperError) as e:
        api.solve_stage(aa, gts[:1], tight, 1, seeds[:1])
    assert e.value.code == cdefs.ERR_SOLVER
    # the per-read assignment behind the optimum: counts of one attempt are one location per read, and the likelihood of that
    # assignment (oracle bookkeeping on the oracle's own model of the genotype) is the optimum
    L, p, loc, aa, ol, oa = setup(gpu_ctx, 4, 12, 3000, seed=17)
    gt = api.generate_genotypes(4, 2)[int(np.argmax(aa.run_filter()))]
    s1 = api.chain_seeds(2, 1)
    lik = api.solve_stage(aa, gt[None, :], exact, 1, s1)[2][0, 0]
    off, counts = api.assignment_counts(aa, gt, exact, 1, s1)
    assert np.all(np.add.reduceat(counts.astype(np.int64), off[:-1].astype(np.int64)) == 1)
    g = O.OracleGtAlns(ol, oa, tuple(int(x) for x in gt))
    g.apply_tweak(int(s1[0]))
    assgn = np.array([int(np.argmax(counts[off[r]:off[r + 1]])) for r in range(len(off) - 1)], dtype=np.uint16)
    assert g.likelihood(assgn)[0] == pytest.approx(lik, rel=1e-9)


def test_exact_at_configs0_size_within_the_reference_solvers_gap(gpu_ctx):
    """BASELINE.json configs[0]: 10 000 read pairs x 8 alleles. The reference's HiGHS run stops — and reports "optimal" — at its default relative
    gap of 1e-4 (highs.rs:103-116 changes no option); that gap is lcty_solver_default's for this kind, and with it the exact solver answers for
    the genotypes of a stage (the best 16 of the prefilter here, one attempt each, solved by the pool of host threads — and, below, all 36
    genotypes of the locus: until round 6 twelve of them were refused, their single-move incumbents 1-2e-4 below the bound), and its
    likelihood is not below what the greedy and the annealing chains of the same attempt reach. Against HiGHS itself on the reference's
    programme: tests/test_exact_highs.py (CPU). The bound behind it: the window counts dualised, multipliers set by subgradient steps at the
    root (1.6e-2 -> ~1e-4 relative at this size)."""
    import time
    L = synth.SynthLocus(8, 10_000, seed=synth.SEED + 3)
    p = api.resolve_params(api.default_params(), L.bg)
    loc = api.Locus(gpu_ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    aa = api.AllAlignments.load(loc, L.reads(0, 10_000))
    gts = api.generate_genotypes(8, 2)
    sub = np.ascontiguousarray(gts[np.argsort(-aa.run_filter(), kind="stable")[:16]])
    seeds = api.chain_seeds(5, len(sub))
    ex = api.default_solver(cdefs.SOLVER_EXACT)
    api.solve_stage(aa, sub[:1], ex, 1, seeds[:1])                      # first use: workspace, depth table
    t0 = time.perf_counter()
    el = api.solve_stage(aa, sub, ex, 1, seeds)[2][:, 0]
    wall = time.perf_counter() - t0
    assert np.all(np.isfinite(el))
    for kind in (cdefs.SOLVER_GREEDY, cdefs.SOLVER_ANNEAL):
        cl = api.solve_stage(aa, sub, api.default_solver(kind), 1, seeds)[2][:, 0]
        assert np.all(cl <= el + 1e-9 * np.abs(el)), (kind, cl, el)
    # the pool: one thread gives the same answers (every model is solved on its own), and takes longer when the machine has cores to spare
    gpu_ctx.set_knob("exact_threads", 1)
    t0 = time.perf_counter()
    e1 = api.solve_stage(aa, sub, ex, 1, seeds)[2][:, 0]
    wall1 = time.perf_counter() - t0
    gpu_ctx.set_knob("exact_threads", -1)
    assert np.array_equal(e1, el)
    print(f"exact, 16 genotypes x 10 000 read pairs: {wall:.2f} s with the pool, {wall1:.2f} s on one thread")
    # a loose check that the pool is a pool (a regression that serialises it, or a host_threads knob that collapses it to one thread,
    # would take as long as one thread does; sixteen models on sixteen or more cores are several times faster): not a benchmark
    import os
    if (os.cpu_count() or 1) >= 16 and wall1 > 0.2:
        assert wall < 1.2 * wall1, f"the exact solver's pool took {wall:.2f} s, one thread {wall1:.2f} s"
    # every genotype of the locus is answered (moves of two reads that meet in a window lift the incumbent into the gap), none below its chains
    every = np.ascontiguousarray(gts[np.argsort(-aa.run_filter(), kind="stable")])
    s36 = api.chain_seeds(6, len(every))
    e36 = api.solve_stage(aa, every, ex, 1, s36)[2][:, 0]
    assert len(e36) == 36 and np.all(np.isfinite(e36))
    for kind in (cdefs.SOLVER_GREEDY, cdefs.SOLVER_ANNEAL):
        c36 = api.solve_stage(aa, every, api.default_solver(kind), 1, s36)[2][:, 0]
        assert np.all(c36 <= e36 + 1e-9 * np.abs(e36)), (kind, c36 - e36)
    # the proof of optimality itself (gap 0) still runs out of nodes at this size: Error::Solver, as a HiGHS run that is not "optimal"
    with pytest.raises(_lib.LocityperError) as e:
        api.solve_stage(aa, sub[:1], proof(), 1, seeds[:1])
    assert e.value.code == cdefs.ERR_SOLVER


def test_attempts_without_a_tweak_share_one_model(gpu_ctx):
    """tweak = 0: apply_tweak draws nothing (assgn.rs:127-151), the attempts of a genotype see one model; it is solved once and every attempt's
    records carry the assignment (the per-read counts of `attempts` attempts are `attempts` x one attempt's)."""
    L = synth.SynthLocus(4, 60, seed=synth.SEED + 11)
    prm = api.default_params(); prm.tweak = 0
    p = api.resolve_params(prm, L.bg)
    assert p.tweak == 0
    loc = api.Locus(gpu_ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    aa = api.AllAlignments.load(loc, L.reads(0, 60))
    gts = api.generate_genotypes(4, 2)[:5]
    ex = api.default_solver(cdefs.SOLVER_EXACT)
    l3 = api.solve_stage(aa, gts, ex, 3, api.chain_seeds(1, 15))[2]
    l1 = api.solve_stage(aa, gts, ex, 1, api.chain_seeds(2, 5))[2]
    assert np.array_equal(l3, np.repeat(l1, 3, axis=1))
    off, c3 = api.assignment_counts(aa, gts[0], ex, 3, api.chain_seeds(1, 3))
    _, c1 = api.assignment_counts(aa, gts[0], ex, 1, api.chain_seeds(1, 1))
    assert np.array_equal(c3, 3 * c1)


def test_exact_through_the_c_abi_against_highs_on_the_references_programme(gpu_ctx):
    """BASELINE configs[0] through the library (the model built on the device from the scored batch, solved by the pool of host threads)
    against HiGHS (scipy.optimize.milp) on `HighsSolver::define_model`'s programme (highs.rs:38-134) built from the oracle's
    GenotypeAlignments with the same tweak: the best genotype of the prefilter and the one of rank 18 (refused until round 6). Both solvers
    stop inside the default relative gap of 1e-4; HiGHS ends at its root node, at the optimum."""
    import ctypes as C
    from tests import pyref_highs as H
    L, p, loc, aa, ol, oa = setup(gpu_ctx, 8, 10_000, 50_000, seed=synth.SEED + 3)
    gts = api.generate_genotypes(8, 2)
    order = np.argsort(-aa.run_filter(), kind="stable")
    ex = api.default_solver(cdefs.SOLVER_EXACT)
    lib = O.lib()
    lib.orc_depth_ln_prob.restype = C.c_double
    for rank in (0, 18):
        gt = np.ascontiguousarray(gts[order[rank]][None, :])
        seed = api.chain_seeds(40 + rank, 1)
        lik = api.solve_stage(aa, gt, ex, 1, seed)[2][0, 0]
        g = O.OracleGtAlns(ol, oa, tuple(int(x) for x in gt[0]))
        g.apply_tweak(int(seed[0]))
        a = g.arrays()
        gc, w = g.window_distr()
        ok, h_assgn, _, info = H.solve(a["read_ixs"], a["ln_prob"], a["windows"], gc, w,
                                       lambda ww, d: lib.orc_depth_ln_prob(ol._h, int(gc[ww]), float(w[ww]), int(d)),
                                       1.0 - p.lik_skew, 1.0 + p.lik_skew, time_limit=300.0)
        assert ok, info
        h_lik = g.likelihood(h_assgn)[0]
        gap = 1e-4 * abs(h_lik)
        assert h_lik - 0.5 * gap <= lik <= h_lik + gap, (rank, lik, h_lik, info)
