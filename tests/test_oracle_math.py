"""Oracle distribution math vs mpmath/scipy golden values (tests/golden/distr_golden.json) and
hand-derived identities. Pins a9/a10/a15/a26/a33 of SURVEY.md §8(a) as far as they can be pinned
without reference vectors."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

from tests import oracle_ffi as O
from locityper_amd import cdefs

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "distr_golden.json")))


def test_ln_gamma_matches_mpmath():
    for x, want in GOLD["ln_gamma"]:
        got = O.lib().orc_ln_gamma(x)
        assert abs(got - want) <= 2e-14 * max(1.0, abs(want)), (x, got, want)


def test_nbinom_ln_pmf_and_cdf():
    L = O.lib()
    for n, p, k, lp, cdf in GOLD["nbinom"]:
        d = L.orc_nbinom_new(n, p)
        got = L.orc_nbinom_ln_pmf(C.byref(d), k)
        assert abs(got - lp) <= 1e-11 * max(1.0, abs(lp)), (n, p, k, got, lp)
        gc = L.orc_nbinom_cdf(C.byref(d), k)
        assert abs(gc - cdf) <= 1e-10, (n, p, k, gc, cdf)


def test_nbinom_mode_and_quantile():
    L = O.lib()
    d = L.orc_nbinom_new(34.0278, 0.0703125)
    mode = L.orc_nbinom_mode(C.byref(d))
    # mode = floor((n-1)(1-p)/p) (nbinom.rs:78-80) and is the argmax of the pmf
    assert mode == math.floor((34.0278 - 1.0) * (1 - 0.0703125) / 0.0703125)
    lp = [L.orc_nbinom_ln_pmf(C.byref(d), k) for k in range(mode - 3, mode + 4)]
    assert int(np.argmax(lp)) == 3
    q = L.orc_nbinom_quantile(C.byref(d), 0.99999)
    k = int(q)
    assert L.orc_nbinom_cdf(C.byref(d), k) <= 0.99999 <= L.orc_nbinom_cdf(C.byref(d), k + 1)
    assert L.orc_insert_cache_size(C.byref(d)) == min(65536, k)
    assert L.orc_nbinom_quantile(C.byref(d), 0.0) == 0.0 and math.isinf(L.orc_nbinom_quantile(C.byref(d), 1.0))


def test_beta_reg():
    for a, b, x, want in GOLD["beta_reg"]:
        got = O.lib().orc_beta_reg(a, b, x)
        assert abs(got - want) <= 1e-12, (a, b, x, got, want)
    assert O.lib().orc_beta_reg(2.0, 3.0, 0.0) == 0.0 and O.lib().orc_beta_reg(2.0, 3.0, 1.0) == 1.0


def test_betabinom_inv_cdf2():
    for a, b, n, c1, c2, k1, k2 in GOLD["betabinom_inv_cdf2"]:
        g1, g2 = C.c_uint32(), C.c_uint32()
        O.lib().orc_betabinom_inv_cdf2(a, b, n, c1, c2, C.byref(g1), C.byref(g2))
        assert (g1.value, g2.value) == (k1, k2), (a, b, n, c1, c2)


def test_students_t_and_t_tests():
    L = O.lib()
    for df, x, want in GOLD["students_t_cdf"]:
        assert abs(L.orc_students_t_cdf(df, x) - want) <= 1e-12
    # unpaired_onesided_t_test::<false> (math/mod.rs:180-198) against scipy's Welch test
    from scipy import stats
    rng = np.random.default_rng(5)
    a, b = rng.normal(-100.0, 3.0, 20), rng.normal(-98.0, 5.0, 20)
    want = stats.ttest_ind(a, b, equal_var=False, alternative="less").pvalue
    got = L.orc_t_test(a.mean(), a.var(ddof=1), b.mean(), b.var(ddof=1), 20.0)
    assert abs(got - want) <= 1e-10
    b2 = rng.normal(-98.0, 5.0, 13)
    want = stats.ttest_ind(a, b2, equal_var=False, alternative="less").pvalue
    got = L.orc_t_test_diffsizes(a.mean(), a.var(ddof=1), b2.mean(), b2.var(ddof=1), 20.0, 13.0)
    assert abs(got - want) <= 1e-10


def test_ln_add_sum():
    L = O.lib()
    ninf = float("-inf")
    assert L.orc_ln_add(ninf, -3.0) == -3.0 and L.orc_ln_add(-3.0, ninf) == -3.0
    assert abs(L.orc_ln_add(math.log(0.2), math.log(0.3)) - math.log(0.5)) < 1e-15
    v = np.log(np.array([0.1, 0.2, 0.05]))
    assert abs(L.orc_ln_sum(v.ctypes.data, 3) - math.log(0.35)) < 1e-15
    assert abs(L.orc_ln_sum_init(v.ctypes.data, 3, math.log(0.15)) - math.log(0.5)) < 1e-15
    assert L.orc_ln_sum(v.ctypes.data, 0) == ninf
    assert L.orc_ln_sum_init(v.ctypes.data, 0, -1.5) == -1.5


def _bg(paired=True):
    bg = cdefs.Bg()
    for i in range(cdefs.GC_BINS):
        bg.depth_n[i], bg.depth_p[i] = 20.0, 2.0 / 3.0
    bg.is_paired = int(paired)
    return bg


def test_bayes_depth_lut():
    p = O.default_params()
    for n, pp, mul, k, want in GOLD["bayes_depth"]:
        bg = _bg(paired=(mul == 2.0))
        bg.depth_n[7], bg.depth_p[7] = n, pp
        got = O.lib().orc_depth_ln_pmf(C.byref(bg), C.byref(p), 7, k)
        assert abs(got - want) <= 1e-9 * max(1.0, abs(want)), (n, pp, mul, k, got, want)


def test_edit_thresholds_fraction_and_pvalue():
    bg = _bg()
    bg.edit_kind, bg.edit_p1, bg.edit_p2 = cdefs.EDIT_FRACTION, 0.03, 0.06
    g, p = C.c_uint32(), C.c_uint32()
    for rl, want in [(150, (4, 9)), (100, (3, 6)), (250, (7, 15)), (33, (0, 1))]:
        O.lib().orc_edit_thresholds(C.byref(bg), rl, C.byref(g), C.byref(p))
        assert (g.value, p.value) == want
    bg.edit_kind, bg.edit_p1, bg.edit_p2 = cdefs.EDIT_PVALUE, 0.99, 0.999
    bg.edit_alpha, bg.edit_beta = 0.6, 90.0
    O.lib().orc_edit_thresholds(C.byref(bg), 150, C.byref(g), C.byref(p))
    assert (g.value, p.value) == (6, 11)


def test_params_defaults_and_resolve():
    p = O.default_params()
    assert (p.boundary_size, p.kmer_soft_thresh, p.kmer_hard_thresh, p.complexity_k) == (200, 5, 1, 5)
    assert p.lik_skew == 0.85 and p.min_weight == 0.001 and p.n_alt_cn == 5
    assert abs(p.filt_diff - 100 * math.log(10)) < 1e-12 and abs(p.prob_thresh + 4 * math.log(10)) < 1e-12
    bg = _bg()
    bg.window, bg.technology = 100, cdefs.TECH_ILLUMINA
    O.resolve_params(p, bg)
    assert p.tweak == 50                                       # min(round(window/2), 200, boundary-1)
    assert abs(p.unmapped_penalty + 10 * math.log(10)) < 1e-12  # model/mod.rs:55-60
    assert abs(p.prob_diff - 11 * math.log(10)) < 1e-12         # genotype.rs:1294-1296
    p2 = O.default_params()
    bg.window, bg.technology = 5000, cdefs.TECH_NANOPORE
    O.resolve_params(p2, bg)
    assert p2.tweak == 199 and abs(p2.unmapped_penalty + 100 * math.log(10)) < 1e-10
    p3 = O.default_params()
    p3.tweak = 250
    with pytest.raises(ValueError):
        O.resolve_params(p3, bg)
