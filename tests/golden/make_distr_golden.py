"""Generates tests/golden/distr_golden.json: known answers for the distribution math on the path,
computed with mpmath at 50 digits (log-gamma based quantities) and scipy.special.betainc (incomplete
beta) — independent of both the oracle's Lanczos ln_gamma / continued fraction and of libm.

Formulas follow the reference: NBinom::ln_pmf (src/math/distr/nbinom.rs:128-131), NBinom::cdf (145-147),
BetaBinomial::inv_cdf2 (src/math/distr/betabinom.rs:74-102), BayesCalc::ln_pmf (src/math/distr/bayes.rs:27-35)
with DistrCache::new's distributions (src/model/distr_cache.rs:61-75), StudentsT cdf as used by
src/math/mod.rs:180-220.  Run:  python tests/golden/make_distr_golden.py
"""
import json
import os

import mpmath as mp
from scipy import special as sps

mp.mp.dps = 50


def ln_gamma(x):
    return mp.loggamma(x)


def nb_ln_pmf(n, p, k):
    n, p = mp.mpf(n), mp.mpf(p)
    return n * mp.log(p) - ln_gamma(n) + ln_gamma(n + k) - ln_gamma(k + 1) + k * mp.log1p(-p)


def nb_cdf(n, p, k):
    # regularised incomplete beta I_p(n, k+1): scipy (Boost) — mpmath's hypergeometric series does not
    # converge for the large parameters used here
    return sps.betainc(float(n), float(k + 1), float(p))


def ln_beta(a, b):
    return ln_gamma(a) + ln_gamma(b) - ln_gamma(a + b)


def bb_inv_cdf2(alpha, beta, n, cdf1, cdf2):
    alpha, beta = mp.mpf(alpha), mp.mpf(beta)
    m = mp.mpf(n)
    const = -mp.log(m + 1) - ln_beta(alpha, beta)

    def pmf(k):
        return mp.exp(-ln_beta(m - k + 1, k + 1) + ln_beta(k + alpha, m - k + beta) + const)
    cdf = pmf(0)
    k1 = n
    i = 0
    for i in range(n):
        cdf += pmf(i + 1)
        if cdf > cdf1:
            k1 = i
            break
    if cdf > cdf2:
        return k1, k1
    for i in range(k1 + 1, n):
        cdf += pmf(i + 1)
        if cdf > cdf2:
            return k1, i
    return k1, n


def t_cdf(df, x):
    df, x = mp.mpf(df), mp.mpf(x)
    h = df / (df + x * x)
    ib = mp.mpf(sps.betainc(float(df / 2), 0.5, float(h))) / 2
    return ib if x <= 0 else 1 - ib


def bayes_ln_pmf(n, p, mul, alts, k):
    null = nb_ln_pmf(n * mul, p, k)
    terms = [null] + [nb_ln_pmf(n * mul * c, p, k) for c in alts]
    mx = max(terms)
    return null - (mx + mp.log(sum(mp.exp(t - mx) for t in terms)))


def main():
    out = {}
    xs = [0.1, 0.5, 1.0, 1.5, 2.0, 3.7, 10.0, 34.03, 100.5, 451.0, 1234.5, 65536.0, 1e6]
    out["ln_gamma"] = [[x, float(ln_gamma(mp.mpf(x)))] for x in xs]
    nb_cases = []
    for n, p in [(12.5, 0.03), (34.0278, 0.0703125), (2000.0, 2.0 / 3.0), (20.0, 0.6667), (0.7, 0.5)]:
        for k in [0, 1, 5, 50, 400, 450, 1000, 5000, 70000]:
            nb_cases.append([n, p, k, float(nb_ln_pmf(n, p, k)), float(nb_cdf(n, p, k))])
    out["nbinom"] = nb_cases
    out["beta_reg"] = [[a, b, x, float(sps.betainc(a, b, x))]
                       for a, b, x in [(2.0, 3.0, 0.4), (34.03, 451.0, 0.07), (0.5, 0.5, 0.3), (10.0, 0.5, 0.9),
                                       (5.0, 5.0, 0.5), (100.0, 2000.0, 0.05)]]
    out["betabinom_inv_cdf2"] = [[a, b, n, c1, c2, *bb_inv_cdf2(a, b, n, c1, c2)]
                                 for a, b, n, c1, c2 in [(0.6, 90.0, 150, 0.99, 0.999), (6.0, 180.0, 3000, 0.99, 0.999),
                                                         (0.5, 100.0, 250, 0.95, 0.99), (2.0, 50.0, 100, 0.5, 0.999999),
                                                         (1.0, 1.0, 10, 0.5, 0.95)]]
    out["students_t_cdf"] = [[df, x, float(t_cdf(df, x))] for df, x in
                             [(19.0, -2.5), (19.0, 0.0), (19.0, 1.3), (3.7, -4.0), (38.0, 5.0), (1.0, 0.5)]]
    alts = [0.3, 2.0, 3.0, 4.0, 5.0]
    out["bayes_depth"] = [[n, p, mul, k, float(bayes_ln_pmf(mp.mpf(n), mp.mpf(p), mul, alts, k))]
                          for n, p, mul in [(20.0, 2.0 / 3.0, 2.0), (2000.0, 2.0 / 3.0, 2.0), (5.5, 0.4, 1.0)]
                          for k in [0, 3, 10, 20, 40, 100, 255, 1000, 4000]]
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "distr_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(json.dumps(out)), "bytes")


if __name__ == "__main__":
    main()
