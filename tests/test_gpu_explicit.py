"""Explicit region weights (--reg-weights) on the device, through the C ABI, against the oracle: window weights, read-pair
weights, pair alignments, likelihood matrix, prefilter, solver chains, the loader's errors. All tests need an MI355X."""
import numpy as np
import pytest

from locityper_amd import _lib, api, cdefs, synth
from locityper_amd.cdefs import ReadsChunk
from tests import oracle_ffi as O
from tests.helpers import make_bg, random_alleles, locus_arrays, compare_gpu_to_oracle
from tests.test_oracle_explicit import bed_lines, columns
from tests.test_gpu_parity import both_loci, check_prefilter, random_pairs

pytestmark = pytest.mark.gpu


def allele_lengths(L, n):
    return [int(L.seq_off[a + 1] - L.seq_off[a]) for a in range(n)]


@pytest.mark.parametrize("n_alleles,tech,rl,n_pairs", [(8, cdefs.TECH_ILLUMINA, 150, 1500), (70, cdefs.TECH_ILLUMINA, 150, 400),
                                                       (12, cdefs.TECH_NANOPORE, 3000, 300)])
def test_explicit_weights_against_the_oracle(gpu_ctx, n_alleles, tech, rl, n_pairs):
    L = synth.SynthLocus(n_alleles, 10_000, technology=tech, read_len=rl, seed=91)
    loc, ol, p = both_loci(gpu_ctx, L)
    ch = L.reads(0, n_pairs)
    plain = api.AllAlignments.load(loc, ch)
    st0, w0, unm0, uk0 = plain.status()
    ww0 = loc.window_weights()
    lines = bed_lines(allele_lengths(L, n_alleles), np.random.default_rng(17), piece=(50, 2500))
    cols = columns(lines)
    loc.set_explicit_weights(*cols)
    assert ol.set_explicit_weights(*cols) == 0

    # window weights: the oracle's value at every window of every allele, bit for bit given the same plain weight
    ww = loc.window_weights()
    assert np.all(ww <= ww0) and np.any(ww < ww0)
    left_padding = (L.bg.neighb - L.bg.window) // 2
    pos_off = 0
    for a in range(n_alleles):
        _, _, _, nw, rs = ol.contig_info(a)
        for j in range(nw):
            ws = rs + j * L.bg.window
            want = O.lib().orc_window_weight(ol._h, a, ws, None)
            got = ww[pos_off + ws - left_padding]
            assert abs(got - want) <= 1e-12 * max(want, 1e-300), (a, j, got, want)
        pos_off += allele_lengths(L, n_alleles)[a] - L.bg.neighb + 1

    aa, oa = api.AllAlignments.load(loc, ch), ol.load(ch)
    M, Mo = compare_gpu_to_oracle(aa, oa)
    st, w, unm, uk = aa.status()
    assert np.array_equal(uk, uk0)                                      # k-mer counts do not depend on the weights
    kept = (st0 == cdefs.READ_GOOD) | (st0 == cdefs.READ_FEW_KMERS)
    assert np.all(w[kept] <= w0[kept] + 1e-15) and np.any(w[kept] < w0[kept]) and np.array_equal(st[~kept], st0[~kept])
    if L.bg.is_paired:
        check_prefilter(aa, Mo, n_alleles, p)

    # the solver stages see both effects (window weights in the depth term, scaled alignment likelihoods)
    if n_alleles <= 12:
        off, pa = aa.pair_alns()
        ol.inject_tables(loc.depth_lut(), ww)
        oa2 = O.alns_from_arrays(n_alleles, st, w, unm, off, pa)
        gts = api.generate_genotypes(n_alleles, 2)[:10]
        seeds = api.chain_seeds(5, 2 * len(gts))
        for kind in (cdefs.SOLVER_GREEDY, cdefs.SOLVER_ANNEAL):
            sv = api.default_solver(kind)
            if kind == cdefs.SOLVER_ANNEAL:
                sv.anneal_steps, sv.plato_size = 2000, 1500
            gl = api.solve_stage(aa, gts, sv, 2, seeds)[2]
            olk = O.solve_stage(ol, oa2, gts, sv, 2, seeds)[2]
            assert np.abs(gl - olk).max() <= 1e-9 * np.abs(olk).max()

    # a file of ones is the same as no file; a second call replaces the first
    n = allele_lengths(L, n_alleles)
    loc.set_explicit_weights(np.arange(n_alleles), np.zeros(n_alleles), n, np.ones(n_alleles))
    assert np.array_equal(loc.window_weights(), ww0)
    a1 = api.AllAlignments.load(loc, ch)
    st1, w1, unm1, _ = a1.status()
    assert np.array_equal(st1, st0) and np.array_equal(w1, w0) and np.array_equal(unm1, unm0)
    assert np.array_equal(a1.best_aln_matrix(), plain.best_aln_matrix())


def test_explicit_weights_adversarial_pairs(gpu_ctx):
    """Several pair alignments per contig (general path of the kernel), unmapped ends (middle = None), zero-weight regions."""
    rng = np.random.default_rng(404)
    alleles = random_alleles(7, 2600, seed=9, snp_rate=0.02)
    bg = make_bg()
    p = api.resolve_params(api.default_params(), bg)
    seqs, seq_off, cflat, cnt_off, _ = locus_arrays(alleles, 25)
    loc = api.Locus(gpu_ctx, seqs, seq_off, cflat, cnt_off, 25, bg, p)
    ol = O.OracleLocus(seqs, seq_off, cflat, cnt_off, 25, bg, p)
    lines = bed_lines([len(a) for a in alleles], rng, piece=(20, 300), values=(0.0, 0.0, 0.05, 0.5, 1.0))
    cols = columns(lines)
    loc.set_explicit_weights(*cols)
    assert ol.set_explicit_weights(*cols) == 0
    ch = ReadsChunk.from_pairs(random_pairs(rng, alleles, 500, True))
    aa, oa = api.AllAlignments.load(loc, ch), ol.load(ch)
    compare_gpu_to_oracle(aa, oa)
    assert len(set(oa.status.tolist())) >= 3 and np.any(np.diff(oa.pa_off.astype(np.int64)) > 7)


def test_explicit_weights_loader_errors(gpu_ctx):
    alleles = random_alleles(3, 1500, seed=21)
    bg = make_bg()
    p = api.resolve_params(api.default_params(), bg)
    seqs, seq_off, cflat, cnt_off, _ = locus_arrays(alleles, 25)
    loc = api.Locus(gpu_ctx, seqs, seq_off, cflat, cnt_off, 25, bg, p)
    ol = O.OracleLocus(seqs, seq_off, cflat, cnt_off, 25, bg, p)
    n = [len(a) for a in alleles]
    ok = [(0, 0, n[0], 1.0), (1, 0, 100, 0.5), (1, 100, n[1], 0.25), (2, 0, n[2], 0.0)]
    ww0 = loc.window_weights()
    cases = [ok[:3], [(0, 0, n[0] - 1, 1.0)] + ok[1:], [(0, 5, n[0], 1.0)] + ok[1:], [(0, 0, n[0], 1.5)] + ok[1:],
             [(0, 0, n[0], float("nan"))] + ok[1:], [(0, 0, n[0] + 1, 1.0)] + ok[1:], [(0, 7, 7, 1.0)] + ok]
    for lines in cases:
        want = ol.set_explicit_weights(*columns(lines))
        assert want != 0
        with pytest.raises(_lib.LocityperError) as e:
            loc.set_explicit_weights(*columns(lines))
        assert e.value.code == want, (lines[0], e.value.code, want)
        assert np.array_equal(loc.window_weights(), ww0)               # a failed call leaves the locus as it was
    loc.set_explicit_weights(*columns(ok + [(9, 0, 10, 0.5)]))         # unknown contig: skipped
    assert ol.set_explicit_weights(*columns(ok + [(9, 0, 10, 0.5)])) == 0
    assert np.all(loc.window_weights() <= ww0)
