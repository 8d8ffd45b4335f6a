"""Oracle solver stages: RNG known answers, GenotypeAlignments against the Python
transliteration, likelihood bookkeeping against brute force, Greedy / SimAnneal sanity, genotype comparison."""
import ctypes as C
import itertools
import math

import numpy as np
import pytest

from locityper_amd import cdefs, synth
from tests import oracle_ffi as O
from tests import pyref


def small_case(n_alleles=6, n_pairs=300, seed=77, base_len=6000):
    L = synth.SynthLocus(n_alleles, n_pairs, seed=seed, base_len=base_len)
    p = O.resolve_params(O.default_params(), L.bg)
    ol = O.OracleLocus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    oa = ol.load(L.reads(0, n_pairs))
    return L, p, ol, oa


def test_xoshiro_known_answers_and_jump():
    r = O.Rng()
    for i, v in enumerate((1, 2, 3, 4)):
        r.s[i] = v
    # first outputs of xoshiro256++ from state {1,2,3,4}: rotl(1+4,23)+1 and rotl(7+6*2^45,23)+7 (by hand)
    assert O.lib().orc_rng_next(C.byref(r)) == 41943041
    assert O.lib().orc_rng_next(C.byref(r)) == 58720359
    # seed_from_u64 = SplitMix64 expansion: first state word of seed 0 is the published SplitMix64(0) output
    r0 = O.rng_from_seed(0)
    assert r0.s[0] == 0xE220A8397B1DCDAF and r0.s[1] == 0x6E789E6AA1B965F4
    # jump is a linear map: it commutes with stepping
    a, b = O.rng_from_seed(42), O.rng_from_seed(42)
    O.lib().orc_rng_jump(C.byref(a)); O.lib().orc_rng_next(C.byref(a))
    O.lib().orc_rng_next(C.byref(b)); O.lib().orc_rng_jump(C.byref(b))
    assert list(a.s) == list(b.s)
    a, b = O.rng_from_seed(7), O.rng_from_seed(7)
    O.lib().orc_rng_long_jump(C.byref(a)); O.lib().orc_rng_jump(C.byref(a))
    O.lib().orc_rng_jump(C.byref(b)); O.lib().orc_rng_long_jump(C.byref(b))
    assert list(a.s) == list(b.s)
    # adaptors
    r = O.rng_from_seed(3)
    vals = [O.lib().orc_rng_below(C.byref(r), 10) for _ in range(2000)]
    assert min(vals) == 0 and max(vals) == 9 and abs(np.mean(vals) - 4.5) < 0.3
    f = [O.lib().orc_rng_f64(C.byref(r)) for _ in range(2000)]
    assert 0.0 <= min(f) and max(f) < 1.0
    assert O.lib().orc_counter_u64(5, 0) != O.lib().orc_counter_u64(5, 1)


def test_weight_calculator_known_values():
    # weight(breakpoint) = 1/2, monotone, 0 at x = 0 and 1 at x = 1 (windows.rs:163-177)
    wc = O.lib().orc_weight_calc
    assert abs(wc(0.2, 4.0, 0.2) - 0.5) < 1e-15 and abs(wc(0.5, 4.0, 0.5) - 0.5) < 1e-15
    assert wc(0.5, 4.0, 0.0) == 0.0 and wc(0.5, 4.0, 1.0) == 1.0
    assert abs(wc(0.5, 4.0, 0.75) - 1.0 / (1.0 + (1.0 / 3.0) ** 4)) < 1e-15


def test_genotype_alignments_against_python_transliteration():
    L, p, ol, oa = small_case()
    good = np.nonzero(oa.status == cdefs.READ_GOOD)[0]
    for ids in [(1, 5), (2, 2), (0, 3), (4, 1)]:
        g = O.OracleGtAlns(ol, oa, ids)
        arr = g.arrays()
        assert g.n_reads == len(good)
        for rp, r in enumerate(good):
            lo, hi = int(oa.pa_off[r]), int(oa.pa_off[r + 1])
            pas = [(float(x["ln_prob"]), int(x["contig"]), int(x["ix1"]), int(x["mid1"]), int(x["ix2"]), int(x["mid2"]))
                   for x in oa.pair_alns[lo:hi]]
            want = pyref.extend_read_gt_alns(pas, float(oa.unmapped_prob[r]), ids, p.prob_diff)
            a, b = int(arr["read_ixs"][rp]), int(arr["read_ixs"][rp + 1])
            got = [(float(arr["ln_prob"][i]), int(arr["contig_ix"][i]), int(arr["mid1"][i]), int(arr["mid2"][i]))
                   for i in range(a, b)]
            assert got == want, (ids, rp)
        assert sorted(arr["non_trivial"].tolist()) == [rp for rp in range(g.n_reads)
                                                       if arr["read_ixs"][rp + 1] - arr["read_ixs"][rp] > 1]
        # windows: total = 2 + sum n_windows; homozygous genotypes list the allele twice (different shifts)
        nw = [ol.contig_info(a)[3] for a in ids]
        assert g.n_windows == 2 + sum(nw)
        # tweak = 0 is deterministic (define_windows_determ)
    p0 = O.resolve_params(O.default_params(), L.bg)
    p0.tweak = 0
    ol0 = O.OracleLocus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p0)
    oa0 = ol0.load(L.reads(0, 300))
    ids = (1, 5)
    g = O.OracleGtAlns(ol0, oa0, ids)
    g.apply_tweak(99)
    arr = g.arrays()
    infos = [ol0.contig_info(a) for a in ids]
    shifts = [2, 2 + infos[0][3]]
    for i in range(g.n_alns):
        cix = int(arr["contig_ix"][i])
        if cix == 0xFF:
            assert arr["windows"][i].tolist() == [0, 0]
        else:
            nwin, rs = infos[cix][3], infos[cix][4]
            want = [pyref.window_ix(rs, nwin, L.bg.window, shifts[cix], int(arr["mid1"][i])),
                    pyref.window_ix(rs, nwin, L.bg.window, shifts[cix], int(arr["mid2"][i]))]
            assert arr["windows"][i].tolist() == want
    gc, w = g.window_distr()
    assert w[0] == 0.0 and w[1] == 0.0
    for j in (0, 3, infos[0][3] - 1):
        wstart = infos[0][4] + j * L.bg.window
        gcv = C.c_uint32()
        ww = O.lib().orc_window_weight(ol0._h, ids[0], wstart, C.byref(gcv))
        assert (w[2 + j] == ww and gc[2 + j] == gcv.value) or (ww < p0.min_weight and w[2 + j] == 0.0)


def test_tweak_is_deterministic_and_bounded():
    L, p, ol, oa = small_case()
    g = O.OracleGtAlns(ol, oa, (1, 5))
    g.apply_tweak(1)
    a1 = g.arrays()["windows"].copy()
    w1 = g.window_distr()[1].copy()
    g.apply_tweak(2)
    a2 = g.arrays()["windows"].copy()
    g.apply_tweak(1)
    assert np.array_equal(g.arrays()["windows"], a1) and np.array_equal(g.window_distr()[1], w1)
    assert not np.array_equal(a1, a2)
    assert a1.max() < g.n_windows
    # the window of a location moves by at most ceil(2*tweak/window) + 1 windows between draws
    d = np.abs(a1.astype(np.int64) - a2.astype(np.int64))
    regular = (a1 >= 2) & (a2 >= 2)
    assert d[regular].max() <= 2 * p.tweak // L.bg.window + 1


def test_depth_lik_diff_bookkeeping():
    # every (w1,w2)->(w3,w4) move changes the depth vector exactly as the multiplicity table says
    for ws in itertools.product(range(3), repeat=4):
        depth = {0: 5, 1: 5, 2: 5}
        depth[ws[0]] -= 1; depth[ws[1]] -= 1; depth[ws[2]] += 1; depth[ws[3]] += 1
        got = {0: 5, 1: 5, 2: 5}
        for w, c in pyref.depth_lik_diff_counts(*ws):
            got[w] += c
        assert got == depth


def brute_force_best(g, arr, max_states=200000):
    n = g.n_reads
    sizes = [int(arr["read_ixs"][r + 1] - arr["read_ixs"][r]) for r in range(n)]
    total = 1
    for s in sizes:
        total *= s
    assert total <= max_states
    best = -math.inf
    for assgn in itertools.product(*[range(s) for s in sizes]):
        lik, _ = g.likelihood(np.array(assgn, dtype=np.uint16))
        best = max(best, lik)
    return best


def test_solvers_against_brute_force_on_tiny_instance():
    L = synth.SynthLocus(4, 40, seed=9, base_len=4000)
    p = O.resolve_params(O.default_params(), L.bg)
    ol = O.OracleLocus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    oa = ol.load(L.reads(0, 9))
    g = O.OracleGtAlns(ol, oa, L.true_genotype)
    g.apply_tweak(5)
    arr = g.arrays()
    best = brute_force_best(g, arr)
    greedy, anneal = O.default_solver(cdefs.SOLVER_GREEDY), O.default_solver(cdefs.SOLVER_ANNEAL)
    lik0, _ = g.likelihood(np.zeros(g.n_reads, dtype=np.uint16))
    got_g = max(g.solve(greedy, s)[0] for s in range(5))
    got_a = max(g.solve(anneal, s)[0] for s in range(5))
    assert lik0 - 1e-9 <= got_g <= best + 1e-9
    assert got_a <= best + 1e-9
    assert abs(got_a - best) < 1e-6          # annealing finds the optimum of a 9-read instance
    # the incrementally maintained likelihood equals a from-scratch recalculation
    for solver in (greedy, anneal):
        lik, assgn, parts = g.solve(solver, 123)
        lik2, parts2 = g.likelihood(assgn)
        assert abs(lik - lik2) <= 1e-9 * max(1.0, abs(lik2))
        assert np.allclose(parts, parts2, rtol=1e-12, atol=1e-9)
    # likelihood = depth_contrib * depth_lik + aln_contrib * aln_lik (assgn.rs:235-237)
    lik, parts = g.likelihood(np.zeros(g.n_reads, dtype=np.uint16))
    assert abs(lik - ((1 + p.lik_skew) * parts[1] + (1 - p.lik_skew) * parts[0])) < 1e-12
    assert abs(parts[0] - g.max_aln_lik()) < 1e-12


def test_stage_and_final_comparison():
    L, p, ol, oa = small_case(n_pairs=400)
    gts = O.generate_genotypes(6, 2)
    greedy = O.default_solver(cdefs.SOLVER_GREEDY)
    attempts = 3
    seeds = np.arange(len(gts) * attempts, dtype=np.uint64) * 7919 + 11
    mean, var, liks = O.solve_stage(ol, oa, gts, greedy, attempts, seeds)
    mean2, var2, liks2 = O.solve_stage(ol, oa, gts, greedy, attempts, seeds)
    assert np.array_equal(liks, liks2)                                   # deterministic given the chain seeds
    assert np.allclose(mean, liks.mean(axis=1)) and np.allclose(var, liks.var(axis=1, ddof=1))
    assert tuple(gts[int(np.argmax(mean))]) == L.true_genotype
    pri = -np.arange(len(gts), dtype=np.float64)
    mean3, _, _ = O.solve_stage(ol, oa, gts, greedy, attempts, seeds, priors=pri)
    assert np.allclose(mean3, mean + pri)
    att = np.full(len(gts), attempts, dtype=np.uint32)
    keep = O.discard_improbable(mean, var, att, np.arange(len(gts)), p.prob_thresh, 4, 1)
    assert 4 <= len(keep) <= len(gts) and keep[0] == np.argmax(mean)
    assert np.all(np.diff(mean[keep[:4]]) <= 0)
    ix, lp, q = O.produce_result(mean, var, att, keep, p.prob_thresh)
    assert ix[0] == np.argmax(mean) and abs(np.logaddexp.reduce(lp)) < 1e-9
    assert abs(q - min(-10.0 * np.logaddexp.reduce(lp[1:]) / math.log(10), 1e9)) < 1e-6
    # compare_two_likelihoods: symmetric inputs give ln(1/2) from the simple normalisation
    c = O.lib().orc_compare_two_likelihoods(-100.0, float("nan"), 1, -100.0, float("nan"), 1)
    assert abs(c - math.log(0.5)) < 1e-12
    c2 = O.lib().orc_compare_two_likelihoods(-100.0, 4.0, 20, -90.0, 4.0, 20)
    assert c2 <= math.log(0.5) and c2 >= math.log(1.0 / (1.0 + math.exp(10.0))) - 1e-12
    # unexplained reads
    n_un = O.lib().orc_count_unexplained(oa._h, np.array(L.true_genotype, dtype=np.uint16).ctypes.data, 2)
    assert 0 <= n_un < oa.n_good // 10


def test_assignment_counts_bookkeeping():
    # update_counts (assgn.rs:374-378): every attempt adds exactly one count per read pair, at its final location
    L, p, ol, oa = small_case(n_pairs=400)
    gt = np.array(L.true_genotype, dtype=np.uint16)
    greedy = O.default_solver(cdefs.SOLVER_GREEDY)
    attempts = 5
    seeds = np.arange(attempts, dtype=np.uint64) * 104729 + 3
    off, counts = O.assignment_counts(ol, oa, gt, greedy, attempts, seeds)
    assert len(off) == oa.n_good + 1 and off[0] == 0 and off[-1] == len(counts)
    per_read = np.add.reduceat(counts.astype(np.int64), off[:-1].astype(np.int64))
    assert np.all(per_read == attempts)
    nw = np.diff(off.astype(np.int64))
    assert nw.min() >= 1 and np.all(counts[off[:-1][nw == 1].astype(np.int64)] == attempts)      # trivial reads never move
    # the same list of locations as the Python transliteration of extend_read_gt_alns
    g = O.GtAlns(ol, oa, gt) if hasattr(O, "GtAlns") else None
    if g is not None:
        assert g.n_alns == len(counts)
    # one attempt == the assignment orc_solve reports
    off1, c1 = O.assignment_counts(ol, oa, gt, greedy, 1, seeds[:1])
    assert np.array_equal(off1, off) and set(np.unique(c1)) <= {0, 1}


def test_threaded_oracle_equals_the_single_thread_one():
    """The CPU baseline of bench.py runs the oracle with the reference's thread structure (locs.rs:1149-1174, solve.rs:1047-1062):
    the same results for any number of threads, including more threads than reads / genotypes and ragged last blocks."""
    L, p, ol, oa = small_case(n_pairs=700)
    chunk = L.reads(0, 700)
    s0 = oa.status
    for threads in (1, 3, 8, 64):
        ob, secs = ol.load_mt(chunk, threads)
        assert ob.n_good == oa.n_good and secs[0] > 0 and secs[1] > 0
        assert np.array_equal(ob.status, s0) and np.array_equal(ob.weight, oa.weight) and np.array_equal(ob.uniq_kmers, oa.uniq_kmers)
        assert np.array_equal(ob.unmapped_prob, oa.unmapped_prob)
        assert np.array_equal(oa.pa_off, ob.pa_off) and oa.pair_alns.tobytes() == ob.pair_alns.tobytes()
    gts = O.generate_genotypes(6, 2)
    seeds = np.arange(len(gts) * 2, dtype=np.uint64) * 104729 + 5
    pri = -0.25 * np.arange(len(gts), dtype=np.float64)
    for kind in (cdefs.SOLVER_GREEDY, cdefs.SOLVER_ANNEAL):
        ref = O.solve_stage(ol, oa, gts, O.default_solver(kind), 2, seeds, priors=pri)
        for threads in (1, 4, 5, 40):
            got = O.solve_stage(ol, oa, gts, O.default_solver(kind), 2, seeds, priors=pri, threads=threads)
            assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(ref, got))


def test_enumeration_is_the_optimum_of_the_ilp_model():
    """the model of highs.rs:38-100 has the assignment of largest ReadAssignment::likelihood as its optimum; the oracle's
    exhaustive enumeration (orc_solve with LCTY_SOLVER_EXACT) against an independent enumeration in Python, and against the chains."""
    L = synth.SynthLocus(4, 40, seed=9, base_len=4000)
    p = O.resolve_params(O.default_params(), L.bg)
    ol = O.OracleLocus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    oa = ol.load(L.reads(0, 11))
    exact = O.default_solver(cdefs.SOLVER_EXACT)
    for gt, seed in (((0, 1), 5), (L.true_genotype, 6), ((2, 2), 7)):
        g = O.OracleGtAlns(ol, oa, gt)
        g.apply_tweak(seed)
        best = brute_force_best(g, g.arrays())
        lik, assgn, parts = g.solve(exact, seed)
        assert lik == pytest.approx(best, rel=1e-12)
        assert g.likelihood(assgn)[0] == pytest.approx(lik, rel=1e-12)
        for kind in (cdefs.SOLVER_GREEDY, cdefs.SOLVER_ANNEAL):
            assert g.solve(O.default_solver(kind), seed)[0] <= lik + 1e-9
    # more assignments than the limit: no answer (NaN), never a wrong one
    exact.node_limit = 4
    g = O.OracleGtAlns(ol, oa, L.true_genotype)
    assert math.isnan(g.solve(exact, 1)[0])
