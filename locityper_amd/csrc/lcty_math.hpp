// lcty_math.hpp — host-side distribution math used to build the device LUTs.
// f64 throughout; lgamma from libm (the reference uses statrs' Lanczos ln_gamma —
// third-party, agrees to ~1e-14 relative; see DESIGN.md "third-party arithmetic").
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "../../include/locityper_hip.h"

namespace lcty {
namespace math {

constexpr double LN10 = 2.302585092994045684;

inline double ln_gamma(double x) {
    int sign = 0;
    return ::lgamma_r(x, &sign);
}
inline double ln_beta(double a, double b) { return ln_gamma(a) + ln_gamma(b) - ln_gamma(a + b); }

// Ln::add (src/math/mod.rs:29-35)
inline double ln_add(double a, double b) {
    const double ninf = -std::numeric_limits<double>::infinity();
    if (a >= b) return b == ninf ? a : b + std::log1p(std::exp(a - b));
    return a == ninf ? b : a + std::log1p(std::exp(b - a));
}

// Ln::map_sum_init (src/math/mod.rs:80-94)
inline double ln_sum_init(const double* v, size_t n, double init) {
    if (n == 0) return init;
    if (n == 1) return ln_add(init, v[0]);
    double m = init;
    for (size_t i = 0; i < n; i++) m = std::fmax(m, v[i]);
    if (std::isinf(m)) return m;
    double s = std::exp(init - m);
    for (size_t i = 0; i < n; i++) s += std::exp(v[i] - m);
    return m + std::log(s);
}

// Regularised incomplete beta (continued fraction, modified Lentz).
inline double beta_reg(double a, double b, double x) {
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return 1.0;
    const double eps = std::numeric_limits<double>::epsilon() / 2;
    const double tiny = std::numeric_limits<double>::min() / eps;
    const double front = std::exp(a * std::log(x) + b * std::log1p(-x) - ln_beta(a, b));
    const bool flip = x >= (a + 1.0) / (a + b + 2.0);
    if (flip) { std::swap(a, b); x = 1.0 - x; }
    double c = 1.0, d = 1.0 - (a + b) * x / (a + 1.0);
    if (std::fabs(d) < tiny) d = tiny;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= 200; m++) {
        const double mm = m, tm = 2.0 * m;
        double num = mm * (b - mm) * x / ((a - 1.0 + tm) * (a + tm));
        d = 1.0 + num * d; if (std::fabs(d) < tiny) d = tiny;
        c = 1.0 + num / c; if (std::fabs(c) < tiny) c = tiny;
        d = 1.0 / d; h *= d * c;
        num = -(a + mm) * (a + b + mm) * x / ((a + tm) * (a + 1.0 + tm));
        d = 1.0 + num * d; if (std::fabs(d) < tiny) d = tiny;
        c = 1.0 + num / c; if (std::fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        const double delta = d * c;
        h *= delta;
        if (std::fabs(delta - 1.0) <= eps) break;
    }
    const double v = front * h / a;
    return flip ? 1.0 - v : v;
}

// NBinom (src/math/distr/nbinom.rs:23-152)
struct NBinom {
    double n = 1.0, p = 0.5, lnq = 0.0, lnpmf_const = 0.0;
    NBinom() = default;
    NBinom(double n_, double p_) : n(n_), p(p_), lnq(std::log1p(-p_)), lnpmf_const(n_ * std::log(p_) - ln_gamma(n_)) {}
    NBinom mul(double coeff) const { return NBinom(n * coeff, p); }
    double ln_pmf(uint32_t k) const {
        const double x = k;
        return lnpmf_const + ln_gamma(n + x) - ln_gamma(x + 1.0) + x * lnq;
    }
    uint32_t mode() const {
        const double v = std::floor((n - 1.0) * (1.0 - p) / p);
        return v > 0.0 ? static_cast<uint32_t>(v) : 0u;
    }
    double mean() const { return n * (1.0 - p) / p; }
    double cdf(uint32_t k) const { return beta_reg(n, static_cast<double>(k) + 1.0, p); }
    // WithQuantile::quantile (src/math/distr/mod.rs:38-75)
    double quantile(double q) const {
        if (q <= 0.0) return 0.0;
        if (q >= 1.0) return std::numeric_limits<double>::infinity();
        int64_t low = 0, high = static_cast<uint32_t>(2.0 * mean());
        while (cdf(static_cast<uint32_t>(high)) < q) { low = high; high = high ? high * 2 : 1; }
        while (high >= low) {
            const int64_t mid = (low + high) / 2;
            if (cdf(static_cast<uint32_t>(mid)) >= q) high = mid - 1; else low = mid + 1;
        }
        if (high < 0) return 0.0;
        const double c0 = cdf(static_cast<uint32_t>(high)), c1 = cdf(static_cast<uint32_t>(high) + 1);
        const double diff = c1 - c0, x = static_cast<double>(high);
        if (diff == 0.0) return x;
        const double r = (q - c0) / diff;
        return x * (1.0 - r) + (x + 1.0) * r;
    }
};

// cache_size (src/bg/insertsz.rs:39-42)
inline size_t insert_cache_size(const NBinom& d) {
    const double q = d.quantile(0.99999);
    return q >= 65536.0 ? size_t(65536) : static_cast<size_t>(q);
}

// BetaBinomial::inv_cdf2 (src/math/distr/betabinom.rs:74-102)
inline void betabinom_inv_cdf2(double alpha, double beta, uint32_t n, double cdf1, double cdf2,
                               uint32_t* k1_out, uint32_t* k2_out) {
    const double m = n;
    const double const_term = -std::log(m + 1.0) - ln_beta(alpha, beta);
    auto inner = [&](double k) { return -ln_beta(m - k + 1.0, k + 1.0) + ln_beta(k + alpha, m - k + beta); };
    double ln_cdf = -ln_beta(m + 1.0, 1.0) + ln_beta(alpha, m + beta) + const_term;
    uint32_t k1 = n;
    for (uint32_t i = 0; i < n; i++) {
        ln_cdf = ln_add(ln_cdf, inner(i + 1.0) + const_term);
        if (std::exp(ln_cdf) > cdf1) { k1 = i; break; }
    }
    if (std::exp(ln_cdf) > cdf2) { *k1_out = k1; *k2_out = k1; return; }
    for (uint32_t i = k1 + 1; i < n; i++) {
        ln_cdf = ln_add(ln_cdf, inner(i + 1.0) + const_term);
        if (std::exp(ln_cdf) > cdf2) { *k1_out = k1; *k2_out = i; return; }
    }
    *k1_out = k1; *k2_out = n;
}

// EditDistCache::get_anew (src/bg/err_prof.rs:434-443)
inline void edit_thresholds(const lcty_bg& bg, uint32_t read_len, uint32_t* good, uint32_t* passable) {
    if (bg.edit_kind == LCTY_EDIT_FRACTION) {
        const double rl = read_len;
        *good = static_cast<uint32_t>(rl * bg.edit_p1);
        *passable = static_cast<uint32_t>(rl * bg.edit_p2);
    } else {
        betabinom_inv_cdf2(bg.edit_alpha, bg.edit_beta, read_len, bg.edit_p1, bg.edit_p2, good, passable);
    }
}

// DistrCache::new + BayesCalc::ln_pmf for one (gc, depth) — src/model/distr_cache.rs:61-75, bayes.rs:27-35
struct DepthDistr {
    NBinom cn1;
    std::vector<NBinom> alts;
    DepthDistr(const lcty_bg& bg, const lcty_params& prm, uint32_t gc) {
        const double mul_coef = bg.is_paired ? 2.0 : 1.0;
        cn1 = NBinom(bg.depth_n[gc], bg.depth_p[gc]).mul(mul_coef);
        for (uint32_t i = 0; i < prm.n_alt_cn; i++) alts.push_back(cn1.mul(prm.alt_cn[i]));
    }
    double ln_pmf(uint32_t k) const {
        const double null_prob = cn1.ln_pmf(k);
        double probs[LCTY_MAX_ALT_CN + 1];
        for (size_t i = 0; i < alts.size(); i++) probs[i] = alts[i].ln_pmf(k);
        return null_prob - ln_sum_init(probs, alts.size(), null_prob);
    }
};

// F64Ext::mean_variance_or_nan (src/ext/vec.rs:74-116): iter().sum() / n, variance over n - 1 (NaN for one value)
inline void mean_variance_or_nan(const double* l, uint32_t n, double* mean_out, double* var_out) {
    double sum = -0.0;
    for (uint32_t a = 0; a < n; a++) sum += l[a];
    const double mean = sum / static_cast<double>(n);
    double var = std::numeric_limits<double>::quiet_NaN();
    if (n > 1) {
        double acc = 0.0;
        for (uint32_t a = 0; a < n; a++) { const double d = l[a] - mean; acc += d * d; }
        var = acc / static_cast<double>(n - 1);
    }
    *mean_out = mean; *var_out = var;
}

}  // namespace math
}  // namespace lcty
