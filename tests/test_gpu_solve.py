"""GPU solver stages (through the C ABI) against the oracle. The oracle is fed the GPU's own tables and
AllAlignments products (test hooks), so every chain sees bit-identical inputs and must follow the same
trajectory; likelihoods then differ only by the order of two big sums."""
import numpy as np
import pytest

from locityper_amd import _lib, api, cdefs, synth
from tests import oracle_ffi as O

pytestmark = pytest.mark.gpu


def setup(ctx, n_alleles, n_pairs, base_len, seed=31, tech=cdefs.TECH_ILLUMINA, read_len=150, **prm):
    L = synth.SynthLocus(n_alleles, n_pairs, seed=seed, base_len=base_len, technology=tech, read_len=read_len)
    p = api.default_params()
    for k, v in prm.items():
        setattr(p, k, v)
    api.resolve_params(p, L.bg)
    loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    aa = api.AllAlignments.load(loc, L.reads(0, n_pairs))
    st, w, unm, uk = aa.status()
    off, pa = aa.pair_alns()
    ol = O.OracleLocus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    ol.inject_tables(loc.depth_lut(), loc.window_weights())
    oa = O.alns_from_arrays(n_alleles, st, w, unm, off, pa)
    return L, p, loc, aa, ol, oa


def compare_stage(aa, ol, oa, gts, solver, attempts, seeds, priors=None):
    gm, gv, gl = api.solve_stage(aa, gts, solver, attempts, seeds, priors)
    om, ov, olk = O.solve_stage(ol, oa, gts, solver, attempts, seeds, priors)
    assert np.abs(gl - olk).max() <= 1e-9 * np.abs(olk).max()            # same trajectories, f64 sums in another order
    assert np.allclose(gm, om, rtol=1e-9) and np.allclose(gv, ov, rtol=1e-6, atol=1e-9, equal_nan=True)
    return gm, gv, gl


def test_window_weight_table(gpu_ctx):
    L, p, loc, aa, ol, oa = setup(gpu_ctx, 6, 500, 8000)
    ol2 = O.OracleLocus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)       # without injection
    ww = loc.window_weights()
    import ctypes as C
    pos = 0
    for a in range(6):
        npos = int(L.seq_off[a + 1] - L.seq_off[a]) - L.bg.neighb + 1
        left = (L.bg.neighb - L.bg.window) // 2
        for i in (0, 7, npos // 2, npos - 1):
            want = O.lib().orc_window_weight(ol2._h, a, i + left, None)
            assert abs(ww[pos + i] - want) <= 1e-12 * max(1.0, want)
        pos += npos
    assert 0.0 <= ww.min() and ww.max() <= 1.0


@pytest.mark.parametrize("n_alleles,n_pairs,base_len,n_gt,attempts", [(8, 3000, 20000, 16, 2), (16, 12000, 30000, 30, 3)])
def test_greedy_and_anneal_chains_match_oracle(gpu_ctx, n_alleles, n_pairs, base_len, n_gt, attempts):
    L, p, loc, aa, ol, oa = setup(gpu_ctx, n_alleles, n_pairs, base_len)
    gts = api.generate_genotypes(n_alleles, 2)
    order = np.argsort(-aa.run_filter())[:n_gt]
    sub = gts[order]
    seeds = api.chain_seeds(2024, len(sub) * attempts)
    for kind in (cdefs.SOLVER_GREEDY, cdefs.SOLVER_ANNEAL):
        solver = api.default_solver(kind)
        gm, gv, gl = compare_stage(aa, ol, oa, sub, solver, attempts, seeds)
        assert tuple(sub[int(np.argmax(gm))]) == L.true_genotype
    # priors shift the likelihood of every attempt (solve.rs:827)
    pri = -np.arange(len(sub), dtype=np.float64)
    g0 = api.default_solver(cdefs.SOLVER_GREEDY)
    m0, _, _ = api.solve_stage(aa, sub, g0, attempts, seeds)
    m1, _, _ = api.solve_stage(aa, sub, g0, attempts, seeds, pri)
    assert np.allclose(m1, m0 + pri, rtol=1e-12)
    # non-default solver parameters: random start, other sample / plateau sizes, short annealing
    g1 = api.default_solver(cdefs.SOLVER_GREEDY)
    g1.best_start, g1.sample_size, g1.plato_size = 0, 3, 30
    compare_stage(aa, ol, oa, sub[:6], g1, attempts, seeds[:6 * attempts])
    a1 = api.default_solver(cdefs.SOLVER_ANNEAL)
    a1.anneal_steps, a1.plato_size, a1.init_prob = 500, 200, 0.3
    compare_stage(aa, ol, oa, sub[:6], a1, attempts, seeds[:6 * attempts])


def test_chain_seeds_are_the_xoshiro_stream(gpu_ctx):
    import ctypes as C
    r = O.rng_from_seed(77)
    want = [O.lib().orc_rng_next(C.byref(r)) for _ in range(5)]
    assert api.chain_seeds(77, 5).tolist() == want


def test_homozygous_single_end_and_deep_coverage(gpu_ctx):
    # homozygous genotypes list the allele twice (duplicate locations, different window shifts)
    L, p, loc, aa, ol, oa = setup(gpu_ctx, 6, 2500, 12000)
    homo = np.array([[a, a] for a in range(6)], dtype=np.uint16)
    seeds = api.chain_seeds(5, 12)
    compare_stage(aa, ol, oa, homo, api.default_solver(cdefs.SOLVER_GREEDY), 2, seeds)
    compare_stage(aa, ol, oa, homo[:3], api.default_solver(cdefs.SOLVER_ANNEAL), 2, seeds[:6])
    # ploidy 1 and 3
    for ploidy in (1, 3, 4):
        g = api.generate_genotypes(6, ploidy)[:5]
        compare_stage(aa, ol, oa, g, api.default_solver(cdefs.SOLVER_GREEDY), 1, api.chain_seeds(9, 5))
    # single-end long reads
    L, p, loc, aa, ol, oa = setup(gpu_ctx, 6, 300, 40000, tech=cdefs.TECH_NANOPORE, read_len=4000)
    g = api.generate_genotypes(6, 2)[:8]
    compare_stage(aa, ol, oa, g, api.default_solver(cdefs.SOLVER_GREEDY), 2, api.chain_seeds(3, 16))
    # depths beyond the 256-entry LinearCache: BayesCalc evaluated directly with the device lgamma
    L, p, loc, aa, ol, oa = setup(gpu_ctx, 4, 20000, 4000)
    g = api.generate_genotypes(4, 2)
    seeds = api.chain_seeds(8, len(g))
    gm, gv, gl = api.solve_stage(aa, g, api.default_solver(cdefs.SOLVER_GREEDY), 1, seeds)
    om, ov, olk = O.solve_stage(ol, oa, g, api.default_solver(cdefs.SOLVER_GREEDY), 1, seeds)
    # lgamma implementations differ by ~1e-13, a near-tie may flip: compare statistically, not per chain
    assert np.abs(gm - om).max() <= 1e-3 * np.abs(om).max()
    assert int(np.argmax(gm)) == int(np.argmax(om))


def test_tweak_zero_and_min_weight(gpu_ctx):
    L, p, loc, aa, ol, oa = setup(gpu_ctx, 6, 2000, 12000, tweak=0)
    g = api.generate_genotypes(6, 2)[:10]
    compare_stage(aa, ol, oa, g, api.default_solver(cdefs.SOLVER_GREEDY), 2, api.chain_seeds(4, 20))


def test_final_comparison_and_full_scheme(gpu_ctx):
    L, p, loc, aa, ol, oa = setup(gpu_ctx, 12, 6000, 20000)
    gts = api.generate_genotypes(12, 2)
    greedy = api.default_solver(cdefs.SOLVER_GREEDY)
    attempts = 3
    seeds = api.chain_seeds(11, len(gts) * attempts)
    gm, gv, gl = api.solve_stage(aa, gts, greedy, attempts, seeds)
    att = np.full(len(gts), attempts, dtype=np.uint32)
    ix = np.arange(len(gts))
    for out_size in (4, 20, 600):
        k1 = api.discard_improbable(gm, gv, att, ix, p.prob_thresh, out_size, 1)
        k2 = O.discard_improbable(gm, gv, att, ix, p.prob_thresh, out_size, 1)
        assert np.array_equal(k1, k2)
    k = api.discard_improbable(gm, gv, att, ix, p.prob_thresh, 20, 1)
    i1, lp1, q1 = api.produce_result(gm, gv, att, k, p.prob_thresh)
    i2, lp2, q2 = O.produce_result(gm, gv, att, k, p.prob_thresh)
    assert np.array_equal(i1, i2) and np.allclose(lp1, lp2, rtol=1e-9, atol=1e-9) and abs(q1 - q2) <= 1e-6 * max(1.0, abs(q2))
    # whole scheme (prefilter -> greedy -> anneal -> result) calls the true genotype
    res = api.solve(aa, p, scheme=(("greedy", 40, 1), ("anneal", 6, 4)), master_seed=3)
    assert tuple(res["genotypes"][0]) == L.true_genotype
    assert abs(np.logaddexp.reduce(res["ln_probs"])) < 1e-9 and res["quality"] >= 0.0
    assert res["kept_per_stage"][0] >= 40
    # output assembly (a34): unexplained reads of the call and of a wrong genotype
    import ctypes as C
    for gt in (res["genotypes"][0], np.array([0, 1], dtype=np.uint16), np.array([3, 3], dtype=np.uint16)):
        g16 = np.ascontiguousarray(gt, dtype=np.uint16)
        assert api.count_unexplained(aa, g16) == O.lib().orc_count_unexplained(oa._h, g16.ctypes.data, 2)
    assert res["unexpl_reads"] == api.count_unexplained(aa, res["genotypes"][0])


def test_solver_misuse_fails_loudly(gpu_ctx):
    L, p, loc, aa, ol, oa = setup(gpu_ctx, 4, 300, 6000)
    g = api.generate_genotypes(4, 2)
    with pytest.raises(_lib.LocityperError):
        api.solve_stage(aa, np.array([[0, 9]], dtype=np.uint16), api.default_solver(0), 1, api.chain_seeds(1, 1))
    with pytest.raises(_lib.LocityperError) as e:
        api.solve_stage(aa, api.generate_genotypes(4, 5)[:2], api.default_solver(0), 1, api.chain_seeds(1, 2))
    assert e.value.code == cdefs.ERR_UNSUPPORTED
    bad = api.default_solver(1)
    bad.init_prob = 0.0
    with pytest.raises(_lib.LocityperError):
        api.solve_stage(aa, g[:1], bad, 1, api.chain_seeds(1, 1))


def test_per_read_assignment_counts_match_oracle(gpu_ctx):
    # the "per-read posteriors" of the output BAMs: counts / attempts (assgn.rs:374-378, model/bam.rs)
    L, p, loc, aa, ol, oa = setup(gpu_ctx, 8, 3000, 20000)
    for gt, kind, attempts in (((L.true_genotype), cdefs.SOLVER_ANNEAL, 6), ((0, 0), cdefs.SOLVER_GREEDY, 3),
                               ((1, 5), cdefs.SOLVER_GREEDY, 4)):
        gt = np.array(gt, dtype=np.uint16)
        solver = api.default_solver(kind)
        seeds = api.chain_seeds(99, attempts)
        off, counts = api.assignment_counts(aa, gt, solver, attempts, seeds)
        ooff, ocounts = O.assignment_counts(ol, oa, gt, solver, attempts, seeds)
        assert np.array_equal(off, ooff)
        assert np.array_equal(counts, ocounts)
        assert np.all(np.add.reduceat(counts.astype(np.int64), off[:-1].astype(np.int64)) == attempts)
    # ploidy 3 and a too-small buffer
    gt3 = np.array([0, 2, 2], dtype=np.uint16)
    off, counts = api.assignment_counts(aa, gt3, api.default_solver(cdefs.SOLVER_GREEDY), 2, api.chain_seeds(5, 2))
    ooff, ocounts = O.assignment_counts(ol, oa, gt3, api.default_solver(cdefs.SOLVER_GREEDY), 2, api.chain_seeds(5, 2))
    assert np.array_equal(off, ooff) and np.array_equal(counts, ocounts)
    import ctypes as C
    n = C.c_uint64()
    small = np.zeros(4, dtype=np.uint16)
    sv = api.default_solver(cdefs.SOLVER_GREEDY)
    seeds = api.chain_seeds(5, 2)
    rc = _lib.lib().lcty_assignment_counts(aa._h, gt3.ctypes.data, 3, C.byref(sv), 2, seeds.ctypes.data, off.ctypes.data,
                                           small.ctypes.data, 4, C.byref(n))
    assert rc == cdefs.ERR_INVALID_INPUT


def test_solver_edge_cases(gpu_ctx):
    """Degenerate inputs the reference handles implicitly: no good reads, one read, every read trivial (Solver::solve
    returns the only assignment, solvers/mod.rs:57-66), fewer non-trivial reads than the greedy sample."""
    from tests.helpers import make_bg, random_alleles, locus_arrays
    from tests.test_gpu_parity import edge_pairs
    from locityper_amd.cdefs import ReadsChunk
    alleles = random_alleles(3, 2600, seed=11)
    bg = make_bg()
    p = api.resolve_params(api.default_params(), bg)
    seqs, seq_off, cflat, cnt_off, _ = locus_arrays(alleles, 25)
    loc = api.Locus(gpu_ctx, seqs, seq_off, cflat, cnt_off, 25, bg, p)
    ol = O.OracleLocus(seqs, seq_off, cflat, cnt_off, 25, bg, p)
    ol.inject_tables(loc.depth_lut(), loc.window_weights())
    pairs = edge_pairs(alleles)
    gts = api.generate_genotypes(3, 2)
    for sub in (pairs, pairs[:1], pairs[:0]):
        ch = ReadsChunk.from_pairs(sub) if sub else None
        if ch is None:
            aa = api.AllAlignments(loc, 1, 64, 1, 1)
            aa.score()
        else:
            aa = api.AllAlignments.load(loc, ch)
        st, w, unm, uk = aa.status()
        off, pa = aa.pair_alns()
        oa = O.alns_from_arrays(3, st, w, unm, off, pa)
        for kind in (cdefs.SOLVER_GREEDY, cdefs.SOLVER_ANNEAL):
            solver = api.default_solver(kind)
            seeds = api.chain_seeds(1, len(gts) * 2)
            gm, gv, gl = api.solve_stage(aa, gts, solver, 2, seeds)
            om, ov, olk = O.solve_stage(ol, oa, gts, solver, 2, seeds)
            assert np.allclose(gl, olk, rtol=1e-9, atol=1e-9), (len(sub), kind)
        for gt in gts[:3]:
            o1, c1 = api.assignment_counts(aa, gt, api.default_solver(cdefs.SOLVER_GREEDY), 2, api.chain_seeds(1, 2))
            o2, c2 = O.assignment_counts(ol, oa, gt, api.default_solver(cdefs.SOLVER_GREEDY), 2, api.chain_seeds(1, 2))
            assert np.array_equal(o1, o2) and np.array_equal(c1, c2)
    # a greedy sample larger than the list of non-trivial reads, and the largest supported one
    L, p2, loc2, aa2, ol2, oa2 = setup(gpu_ctx, 4, 60, 5000)
    g = api.generate_genotypes(4, 2)
    big = api.default_solver(cdefs.SOLVER_GREEDY)
    for ss in (64, 33, 1):
        big.sample_size = ss
        compare_stage(aa2, ol2, oa2, g, big, 2, api.chain_seeds(2, 2 * len(g)))


@pytest.mark.parametrize("prm", [
    dict(lik_skew=0.0), dict(lik_skew=0.5, tweak=10), dict(prob_diff=3.0), dict(prob_diff=60.0, tweak=100),
    dict(min_weight=0.5), dict(n_alt_cn=0), dict(n_alt_cn=1), dict(n_alt_cn=2, lik_skew=-0.3),
])
def test_solver_parameter_sweep(gpu_ctx, prm):
    """model::Params that steer the solver stages (model/mod.rs:64-135): likelihood skew, location threshold,
    tweak size, minimal window weight, alternative copy-number hypotheses of the depth distribution."""
    L, p, loc, aa, ol, oa = setup(gpu_ctx, 8, 2500, 15000, seed=5, **prm)
    gts = api.generate_genotypes(8, 2)
    sub = gts[np.argsort(-aa.run_filter(), kind="stable")[:10]]
    seeds = api.chain_seeds(77, 20)
    for kind in (cdefs.SOLVER_GREEDY, cdefs.SOLVER_ANNEAL):
        solver = api.default_solver(kind)
        if kind == cdefs.SOLVER_ANNEAL:
            solver.anneal_steps, solver.plato_size = 3000, 2000
        compare_stage(aa, ol, oa, sub, solver, 2, seeds)
    off, counts = api.assignment_counts(aa, sub[0], api.default_solver(cdefs.SOLVER_GREEDY), 2, seeds[:2])
    ooff, ocounts = O.assignment_counts(ol, oa, sub[0], api.default_solver(cdefs.SOLVER_GREEDY), 2, seeds[:2])
    assert np.array_equal(off, ooff) and np.array_equal(counts, ocounts)


def test_depth_table_widening_gives_the_same_chains(gpu_ctx):
    """A chain that runs past the depth table raises a flag and the batch is repeated with a wider table: the result
    must not depend on where the table started."""
    gts = api.generate_genotypes(4, 2)
    seeds = api.chain_seeds(8, 2 * len(gts))
    out = []
    for start in (-1, 256):
        gpu_ctx.set_knob("depth_table_start", start)
        try:
            L, p, loc, aa, ol, oa = setup(gpu_ctx, 4, 20000, 4000)          # ~600 mates per window: far beyond 256
            res = [api.solve_stage(aa, gts, api.default_solver(k), 2, seeds)[2] for k in (cdefs.SOLVER_GREEDY, cdefs.SOLVER_ANNEAL)]
        finally:
            gpu_ctx.set_knob("depth_table_start", -1)
        out.append(res)
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_chain_batches_do_not_change_results(gpu_ctx):
    """Stages whose per-chain state exceeds the memory budget run in several batches of genotypes."""
    L, p, loc, aa, ol, oa = setup(gpu_ctx, 8, 3000, 20000)
    gts = api.generate_genotypes(8, 2)
    seeds = api.chain_seeds(21, 3 * len(gts))
    pri = -0.5 * np.arange(len(gts), dtype=np.float64)
    one = api.solve_stage(aa, gts, api.default_solver(cdefs.SOLVER_GREEDY), 3, seeds, pri)
    gpu_ctx.set_knob("solve_budget_mb", 1)                               # a few genotypes per batch
    try:
        many = api.solve_stage(aa, gts, api.default_solver(cdefs.SOLVER_GREEDY), 3, seeds, pri)
    finally:
        gpu_ctx.set_knob("solve_budget_mb", -1)
    assert np.array_equal(one[2], many[2]) and np.array_equal(one[0], many[0])
    assert api.solve_stats(aa)[0] == 3 * len(gts)


def test_the_grouping_of_a_stage_does_not_change_a_chain(gpu_ctx):
    """The initialisation of a diploid stage builds the records of several chains at once (groups of chains on shared rows of the location
    table, solve_init_tile_kernel): what a chain starts from — and therefore where it ends — must not depend on which other genotypes are in
    the stage, on their order, or on how many attempts share a group. Stages of every kind of group: all 136 genotypes of 16 alleles (full
    tiles and their edges, homozygous genotypes, a tail of single chains), the same genotypes one by one, in reverse order, and with more
    attempts than a group holds; reads with several pair-alignments on a contig (the deferred path) are in the batch."""
    L, p, loc, aa, ol, oa = setup(gpu_ctx, 16, 4000, 12000)
    gts = api.generate_genotypes(16, 2)
    for kind in (cdefs.SOLVER_GREEDY, cdefs.SOLVER_ANNEAL):
        sv = api.default_solver(kind)
        if kind == cdefs.SOLVER_GREEDY: sv.plato_size = 20
        else: sv.anneal_steps, sv.plato_size = 300, 200
        seeds = api.chain_seeds(5, len(gts))
        whole = api.solve_stage(aa, gts, sv, 1, seeds)[2][:, 0]
        alone = np.array([api.solve_stage(aa, gts[i:i + 1], sv, 1, seeds[i:i + 1])[2][0, 0] for i in range(0, len(gts), 7)])
        assert np.array_equal(whole[::7], alone)
        back = api.solve_stage(aa, np.ascontiguousarray(gts[::-1]), sv, 1, np.ascontiguousarray(seeds[::-1]))[2][:, 0]
        assert np.array_equal(back[::-1], whole)
        # eleven attempts of three genotypes: groups of eight and of three chains of one genotype; attempt 0 carries the seed it had above
        sub = np.ascontiguousarray(gts[[3, 40, 135]])
        s11 = api.chain_seeds(6, 33)
        many = api.solve_stage(aa, sub, sv, 11, s11)[2]
        for g in range(3):
            for a in (0, 7, 10):
                one = api.solve_stage(aa, sub[g:g + 1], sv, 1, s11[g * 11 + a:g * 11 + a + 1])[2][0, 0]
                assert one == many[g, a]
    compare_stage(aa, ol, oa, gts[:24], api.default_solver(cdefs.SOLVER_GREEDY), 2, api.chain_seeds(9, 48))


def test_library_scheme_driver_equals_the_composed_calls(gpu_ctx):
    """lcty_solve (solve::solve in C++ inside the library) against the same sequence of C-ABI calls made from Python."""
    L, p, loc, aa, ol, oa = setup(gpu_ctx, 12, 6000, 20000)
    stages = (cdefs.Stage * 2)()
    stages[0].solver = api.default_solver(cdefs.SOLVER_GREEDY); stages[0].in_size = 40; stages[0].attempts = 1
    stages[1].solver = api.default_solver(cdefs.SOLVER_ANNEAL); stages[1].in_size = 6; stages[1].attempts = 4
    pri = -0.01 * np.arange(api.count_genotypes(12, 2), dtype=np.float64)
    call, mean, var, att = api.solve_locus(aa, stages, master_seed=3, priors=pri)
    ref = api.solve(aa, p, scheme=(("greedy", 40, 1), ("anneal", 6, 4)), master_seed=3, priors=pri)
    n = int(call.n_out)
    assert n == len(ref["ixs"]) and list(call.ixs[:n]) == ref["ixs"].tolist()
    assert np.array_equal(np.array(call.ln_probs[:n]), ref["ln_probs"]) and call.quality == ref["quality"]
    assert np.array_equal(mean, ref["lik_mean"], equal_nan=True) and np.array_equal(att, ref["attempts"])
    assert call.unexpl_reads == ref["unexpl_reads"] and call.n_good == aa.n_good() and call.warnings == 0
    assert call.kept_after_filter == ref["kept_per_stage"][0]
    gts = api.generate_genotypes(12, 2)
    assert tuple(gts[int(call.ixs[0])]) == L.true_genotype
    # the default scheme runs too (fewer genotypes than the first stage takes: no filter, greedy stage on all 78)
    call2, _, _, att2 = api.solve_locus(aa)
    assert call2.kept_after_filter == 78 and tuple(gts[int(call2.ixs[0])]) == L.true_genotype and att2.max() == 20


def test_chains_per_wavefront_and_wide_samples_do_not_change_results(gpu_ctx):
    """The greedy kernel puts 64 / LPC chains into a wavefront (LPC = 16 lanes per chain for samples of up to 16 reads, 32 and 64
    beyond; rows of 12 or 10 lanes — five or six chains — when a stage has more chains than four per SIMD of the device): every
    layout must give the chains of the oracle, including a last wavefront with spare rows and the spare lanes behind the last row."""
    L, p, loc, aa, ol, oa = setup(gpu_ctx, 8, 4000, 20000)
    gts = api.generate_genotypes(8, 2)[:13]                              # 13 chains x 2 attempts = 26: not a multiple of 4
    seeds = api.chain_seeds(77, 26)
    g = api.default_solver(cdefs.SOLVER_GREEDY)
    ref = compare_stage(aa, ol, oa, gts, g, 2, seeds)
    for cpw in (1, 2, 4, 5, 6):
        gpu_ctx.set_knob("solve_chains_per_wave", cpw)
        try:
            got = api.solve_stage(aa, gts, g, 2, seeds)
        finally:
            gpu_ctx.set_knob("solve_chains_per_wave", -1)
        # the same moves; a chain's likelihood is summed over the lanes that applied them, in an order that depends on the layout
        assert np.allclose(got[2], ref[2], rtol=1e-12, atol=0), cpw
    for sample in (1, 16, 17, 32, 40, 64):
        g2 = api.default_solver(cdefs.SOLVER_GREEDY)
        g2.sample_size = sample
        compare_stage(aa, ol, oa, gts[:5], g2, 2, seeds[:10])
    # window weights from the LDS tables (default) and gathered from the chains' rows: the same arithmetic, the same bits
    gpu_ctx.set_knob("solve_lds_weights", 0)
    try:
        gathered = api.solve_stage(aa, gts, g, 2, seeds)
    finally:
        gpu_ctx.set_knob("solve_lds_weights", -1)
    assert np.array_equal(gathered[2], api.solve_stage(aa, gts, g, 2, seeds)[2])
    # annealing with its window weights gathered, in LDS as they are, and as table indices + tables in LDS (the default): the same chains
    a = api.default_solver(cdefs.SOLVER_ANNEAL)
    a.anneal_steps, a.plato_size = 3000, 2000
    ref_a = compare_stage(aa, ol, oa, gts[:6], a, 2, seeds[:12])
    for mode in (0, 1, 2):
        gpu_ctx.set_knob("anneal_lds_weights", mode)
        try:
            assert np.array_equal(api.solve_stage(aa, gts[:6], a, 2, seeds[:12])[2], ref_a[2]), mode
        finally:
            gpu_ctx.set_knob("anneal_lds_weights", -1)
    # rows of 12 and 10 lanes against the oracle directly, with samples that fill them and samples that do not
    for cpw, sample in ((5, 12), (5, 7), (6, 10), (6, 3), (6, 12)):          # the last: a sample of 12 does not fit a row of 10 -> rows of 16
        g3 = api.default_solver(cdefs.SOLVER_GREEDY)
        g3.sample_size = sample
        gpu_ctx.set_knob("solve_chains_per_wave", cpw)
        try:
            compare_stage(aa, ol, oa, gts, g3, 2, seeds)
        finally:
            gpu_ctx.set_knob("solve_chains_per_wave", -1)


def test_reads_with_many_locations_and_the_growth_of_their_runs(gpu_ctx):
    """Locations beyond the second of a read (ploidy > 2; "both unmapped" within reach with a small unmapped penalty) live in a run per
    chain that starts small and grows on demand: same chains as the oracle, whatever the first size."""
    L, p, loc, aa, ol, oa = setup(gpu_ctx, 6, 2500, 12000)
    g3 = api.generate_genotypes(6, 3)[:9]
    seeds = api.chain_seeds(4, 18)
    ref = [compare_stage(aa, ol, oa, g3, api.default_solver(k), 2, seeds)[2] for k in (cdefs.SOLVER_GREEDY, cdefs.SOLVER_ANNEAL)]
    gpu_ctx.set_knob("solve_extra_start", 7)
    try:
        gpu_ctx.trim()                                                   # forget the run size of the stage before
        got = [api.solve_stage(aa, g3, api.default_solver(k), 2, seeds)[2] for k in (cdefs.SOLVER_GREEDY, cdefs.SOLVER_ANNEAL)]
    finally:
        gpu_ctx.set_knob("solve_extra_start", -1)
        gpu_ctx.trim()
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    # ... and when the grown runs leave room for fewer chains than the batch holds, the batch starts again in smaller batches (it used to
    # end the call with "device memory"): a budget in which the nine genotypes fit with runs of 7 entries and no longer with ~2 600
    gpu_ctx.set_knob("solve_extra_start", 7)
    gpu_ctx.set_knob("solve_budget_mb", 2)
    try:
        gpu_ctx.trim()
        tight = [api.solve_stage(aa, g3, api.default_solver(k), 2, seeds)[2] for k in (cdefs.SOLVER_GREEDY, cdefs.SOLVER_ANNEAL)]
    finally:
        gpu_ctx.set_knob("solve_extra_start", -1)
        gpu_ctx.set_knob("solve_budget_mb", -1)
        gpu_ctx.trim()
    assert np.array_equal(tight[0], ref[0]) and np.array_equal(tight[1], ref[1])
    g4 = api.generate_genotypes(6, 4)[:4]
    compare_stage(aa, ol, oa, g4, api.default_solver(cdefs.SOLVER_ANNEAL), 2, api.chain_seeds(6, 8))


def test_queue_of_loci_equals_one_locus_at_a_time(gpu_ctx):
    """lcty_solve_queue overlaps the last stage of a locus (side stream, second host thread) with the head of the next one: every
    entry must get exactly what lcty_solve gives it alone, also when a batch comes back later in the queue."""
    stages = (cdefs.Stage * 2)()
    stages[0].solver = api.default_solver(cdefs.SOLVER_GREEDY); stages[0].in_size = 30; stages[0].attempts = 1
    stages[1].solver = api.default_solver(cdefs.SOLVER_ANNEAL); stages[1].in_size = 5; stages[1].attempts = 4
    cases = [setup(gpu_ctx, 10, 3000 + 700 * i, 15000, seed=40 + i) for i in range(3)]
    batches = [c[3] for c in cases]
    alone = [api.solve_locus(b, stages, master_seed=11 + i)[0] for i, b in enumerate(batches)]
    order = [0, 1, 2, 0, 1]
    calls = api.solve_queue([batches[i] for i in order], stages, master_seeds=[11 + i for i in order])
    for i, c in zip(order, calls):
        a = alone[i]
        n = int(a.n_out)
        assert int(c.n_out) == n and list(c.ixs[:n]) == list(a.ixs[:n])
        assert list(c.ln_probs[:n]) == list(a.ln_probs[:n]) and c.quality == a.quality
        assert (c.unexpl_reads, c.n_good, c.warnings, c.kept_after_filter) == (a.unexpl_reads, a.n_good, a.warnings, a.kept_after_filter)
        assert tuple(api.generate_genotypes(10, 2)[int(c.ixs[0])]) == cases[i][0].true_genotype
    # the same queue with the head of a locus (scores, cut, location table) made on the main stream after the chains of the locus before,
    # instead of beside them on the fore stream: nothing but the order of issue differs
    gpu_ctx.set_knob("queue_early_head", 0)
    try:
        late = api.solve_queue([batches[i] for i in order], stages, master_seeds=[11 + i for i in order])
    finally:
        gpu_ctx.set_knob("queue_early_head", -1)
    for c, l in zip(calls, late):
        n = int(c.n_out)
        assert int(l.n_out) == n and list(l.ixs[:n]) == list(c.ixs[:n]) and list(l.ln_probs[:n]) == list(c.ln_probs[:n])
        assert (l.unexpl_reads, l.n_good, l.warnings, l.kept_after_filter) == (c.unexpl_reads, c.n_good, c.warnings, c.kept_after_filter)
    # a queue of two positions (one head beside chains, no tail before it) and of one
    two = api.solve_queue([batches[2], batches[0]], stages, master_seeds=[13, 11])
    assert [list(t.ixs[:int(t.n_out)]) for t in two] == [list(alone[2].ixs[:int(alone[2].n_out)]), list(alone[0].ixs[:int(alone[0].n_out)])]
    one = api.solve_queue([batches[1]], stages, master_seeds=[12])
    assert list(one[0].ln_probs[:int(one[0].n_out)]) == list(alone[1].ln_probs[:int(alone[1].n_out)])
    with pytest.raises(_lib.LocityperError) as e:                       # neighbours must be different loci
        api.solve_queue([batches[0], batches[0]], stages)
    assert e.value.code == cdefs.ERR_INVALID_INPUT


@pytest.mark.parametrize("scheme", ["greedy-greedy-anneal", "anneal only", "greedy only"])
def test_queue_with_other_schemes_equals_one_locus_at_a_time(gpu_ctx, scheme):
    """The queue's three lanes (head of the next locus on the fore stream, chains on the main stream, last stage on the side stream) and the
    two launch gates with schemes of three stages (two of them in the head's chains) and of ONE stage (no chains in the head at all: the
    only stage is the tail's): every entry gets what lcty_solve gives it alone."""
    kinds = {"greedy-greedy-anneal": [(cdefs.SOLVER_GREEDY, 30, 1), (cdefs.SOLVER_GREEDY, 10, 2), (cdefs.SOLVER_ANNEAL, 4, 3)],
             "anneal only": [(cdefs.SOLVER_ANNEAL, 12, 2)], "greedy only": [(cdefs.SOLVER_GREEDY, 20, 2)]}[scheme]
    stages = (cdefs.Stage * len(kinds))()
    for st, (kind, in_size, attempts) in zip(stages, kinds):
        st.solver = api.default_solver(kind); st.in_size = in_size; st.attempts = attempts
    cases = [setup(gpu_ctx, 8, 2500 + 600 * i, 12000, seed=140 + i) for i in range(3)]
    batches = [c[3] for c in cases]
    alone = [api.solve_locus(b, stages, master_seed=31 + i)[0] for i, b in enumerate(batches)]
    order = [0, 1, 2, 1]
    calls = api.solve_queue([batches[i] for i in order], stages, master_seeds=[31 + i for i in order])
    for i, c in zip(order, calls):
        a = alone[i]
        n = int(a.n_out)
        assert int(c.n_out) == n and list(c.ixs[:n]) == list(a.ixs[:n]) and list(c.ln_probs[:n]) == list(a.ln_probs[:n])
        assert (c.unexpl_reads, c.n_good, c.warnings, c.kept_after_filter, c.quality) == (a.unexpl_reads, a.n_good, a.warnings, a.kept_after_filter, a.quality)


def test_fed_queue_of_distinct_loci_equals_the_resident_queue(gpu_ctx):
    """lcty_solve_queue_fed + lcty_reads_reset: five positions over three loci through THREE rotating batch objects, each position uploaded
    (counted alignments, from a loader thread, on the copy stream) while the position before it is solved. Every call equals what the
    locus gets alone; a batch is released before the position three further on is acquired; a source without a batch ends the queue."""
    import threading
    stages = (cdefs.Stage * 2)()
    stages[0].solver = api.default_solver(cdefs.SOLVER_GREEDY); stages[0].in_size = 30; stages[0].attempts = 1
    stages[1].solver = api.default_solver(cdefs.SOLVER_ANNEAL); stages[1].in_size = 5; stages[1].attempts = 4
    loci = []
    for i in range(3):
        L = synth.SynthLocus(10, 3000 + 500 * i, seed=60 + i, base_len=15000)
        p = api.resolve_params(api.default_params(), L.bg)
        loc = api.Locus(gpu_ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
        ch = L.reads(0, 3000 + 500 * i)
        half = ch.n_pairs // 2
        parts = [ch.slice(0, half), ch.slice(half, ch.n_pairs)]
        loci.append((L, loc, [(gpu_ctx.pinned_chunk(c), gpu_ctx.pinned_like(c.counted(loc.allele_len))) for c in parts], ch))
    alone = []
    for i, (L, loc, parts, ch) in enumerate(loci):
        aa = api.AllAlignments.load(loc, ch, counted=True)
        alone.append(api.solve_locus(aa, stages, master_seed=21 + i)[0])
    cap_b = max(c.n_bases for _, _, _, c in loci) + 64
    cap_r = max(len(c.recs) for _, _, _, c in loci)
    rot = [api.AllAlignments(loci[0][1], 4000, cap_b // 32 * 32 + 32, cap_r, 0) for _ in range(3)]
    order = [0, 1, 2, 0, 1]
    ready = [threading.Event() for _ in order]
    free = [threading.Semaphore(1) for _ in range(3)]
    events = []

    def loader():
        for q, j in enumerate(order):
            free[q % 3].acquire()
            events.append(("load", q))
            b = rot[q % 3]
            b.reset(loci[j][1])
            for pc, alns in loci[j][2]:
                b.append(pc, counted=alns)
            ready[q].set()

    def acquire(q):
        ready[q].wait()
        events.append(("acquire", q))
        return rot[q % 3]

    def release(q):
        events.append(("release", q))
        free[q % 3].release()

    th = threading.Thread(target=loader)
    th.start()
    calls = api.solve_queue_fed(len(order), acquire, release, stages, master_seeds=[21 + j for j in order])
    th.join()
    for j, c in zip(order, calls):
        a = alone[j]
        n = int(a.n_out)
        assert int(c.n_out) == n and list(c.ixs[:n]) == list(a.ixs[:n]) and list(c.ln_probs[:n]) == list(a.ln_probs[:n])
        assert (c.unexpl_reads, c.n_good, c.warnings, c.kept_after_filter) == (a.unexpl_reads, a.n_good, a.warnings, a.kept_after_filter)
    # three objects carry the queue: a position is released before the position three further on is acquired (the head of position q + 2 is
    # made beside the chains of position q + 1, i.e. it is acquired before position q — whose last stage has ended by then — is released)
    for q in range(len(order) - 3):
        assert events.index(("release", q)) < events.index(("acquire", q + 3))
    assert [e for e in events if e[0] == "release"] == [("release", q) for q in range(len(order))]
    assert [e for e in events if e[0] == "acquire"] == [("acquire", q) for q in range(len(order))]
    # and with the head of a position made after the chains of the position before it (the order of round 4): two further on
    gpu_ctx.set_knob("queue_early_head", 0)
    try:
        events.clear()
        ready2 = [threading.Event() for _ in order]
        ready[:] = ready2
        for f in free:
            while f.acquire(blocking=False): pass
            f.release()
        th = threading.Thread(target=loader)
        th.start()
        late = api.solve_queue_fed(len(order), acquire, release, stages, master_seeds=[21 + j for j in order])
        th.join()
    finally:
        gpu_ctx.set_knob("queue_early_head", -1)
    for c, l in zip(calls, late):
        n = int(c.n_out)
        assert int(l.n_out) == n and list(l.ixs[:n]) == list(c.ixs[:n]) and list(l.ln_probs[:n]) == list(c.ln_probs[:n])
    for q in range(len(order) - 2):
        assert events.index(("release", q)) < events.index(("acquire", q + 2))
    # a source that has nothing for a position: the queue ends with an error, the positions before it were released
    with pytest.raises(_lib.LocityperError) as e:
        api.solve_queue_fed(2, lambda q: rot[0] if q == 0 else None, None, stages)
    # a batch made for a smaller locus is refused by reset
    L2 = synth.SynthLocus(12, 100, seed=3, base_len=15000)
    loc2 = api.Locus(gpu_ctx, L2.seqs, L2.seq_off, L2.counts, L2.cnt_off, L2.k, L2.bg, api.resolve_params(api.default_params(), L2.bg))
    with pytest.raises(_lib.LocityperError) as e:
        rot[0].reset(loc2)
    assert e.value.code == cdefs.ERR_INVALID_INPUT
