/*
 * lcty_oracle.c — CPU restatement (plain C, f64) of the Locityper scoring + prefilter path.
 * TEST INFRASTRUCTURE ONLY — see lcty_oracle.h. PARITY UNPINNED (no reference vectors exist).
 *
 * Citations are paths relative to the reference crate root (tprodanov/locityper v1.7.2).
 * The code is a restatement written from the reference's behaviour, not a copy: data
 * structures are flat C arrays, the control flow follows the cited lines one to one so
 * that every ordering / tie / rounding decision of the CPU path is reproduced.
 */
#include "lcty_oracle.h"
#include "lcty_oracle_internal.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define MIN(a, b) ((a) < (b) ? (a) : (b))
#define MAX(a, b) ((a) > (b) ? (a) : (b))

/* ======================================================================== */
/* math                                                                      */
/* ======================================================================== */

/* statrs 0.19 src/function/gamma.rs (Lanczos approximation, port of Math.NET):
 * published constants GAMMA_R and GAMMA_DK. */
static const double GAMMA_R = 10.900511;
static const double GAMMA_DK[11] = {
    2.48574089138753565546e-5,  1.05142378581721974210,   -3.45687097222016235469,
    4.51227709466894823700,     -2.98285225323576655721,  1.05639711577126713077,
    -1.95428773191645869583e-1, 1.70970543404441224307e-2, -5.71926117404305781283e-4,
    4.63399473359905636708e-6,  -2.71994908488607703910e-9,
};
static const double LN_PI = 1.1447298858494001741434273513530587116472948129153;
static const double LN_2_SQRT_E_OVER_PI = 0.6207822376352452223455184457816472122518527279025978;
static const double ORC_E = 2.71828182845904523536028747135266250;

double orc_ln_gamma(double x) {
    if (x < 0.5) {
        double s = GAMMA_DK[0];
        for (int i = 1; i < 11; i++) s += GAMMA_DK[i] / ((double)i - x);
        return LN_PI - log(sin(M_PI * x)) - log(s) - LN_2_SQRT_E_OVER_PI
               - (0.5 - x) * log((0.5 - x + GAMMA_R) / ORC_E);
    } else {
        double s = GAMMA_DK[0];
        for (int i = 1; i < 11; i++) s += GAMMA_DK[i] / (x + (double)i - 1.0);
        return log(s) + LN_2_SQRT_E_OVER_PI + (x - 0.5) * log((x - 0.5 + GAMMA_R) / ORC_E);
    }
}

double orc_ln_beta(double a, double b) {
    return orc_ln_gamma(a) + orc_ln_gamma(b) - orc_ln_gamma(a + b);
}

/* Regularised incomplete beta I_x(a,b); statrs function::beta::beta_reg uses the same
 * modified-Lentz continued fraction with the symmetry transform. */
double orc_beta_reg(double a, double b, double x) {
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return 1.0;
    const double eps = 1.1102230246251565e-16;
    const double fpmin = 2.2250738585072014e-308 / eps;
    double bt = exp(orc_ln_gamma(a + b) - orc_ln_gamma(a) - orc_ln_gamma(b) + a * log(x) + b * log1p(-x));
    int symm = x >= (a + 1.0) / (a + b + 2.0);
    if (symm) { double t = a; a = b; b = t; x = 1.0 - x; }
    double qab = a + b, qap = a + 1.0, qam = a - 1.0;
    double c = 1.0, d = 1.0 - qab * x / qap;
    if (fabs(d) < fpmin) d = fpmin;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= 140; m++) {
        double mf = (double)m, m2 = 2.0 * mf;
        double aa = mf * (b - mf) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d; if (fabs(d) < fpmin) d = fpmin;
        c = 1.0 + aa / c; if (fabs(c) < fpmin) c = fpmin;
        d = 1.0 / d; h *= d * c;
        aa = -(a + mf) * (qab + mf) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d; if (fabs(d) < fpmin) d = fpmin;
        c = 1.0 + aa / c; if (fabs(c) < fpmin) c = fpmin;
        d = 1.0 / d;
        double del = d * c;
        h *= del;
        if (fabs(del - 1.0) <= eps) break;
    }
    return symm ? 1.0 - bt * h / a : bt * h / a;
}

/* Ln::add — src/math/mod.rs:29-35 */
double orc_ln_add(double a, double b) {
    if (a >= b) {
        return (b == -INFINITY) ? a : b + log1p(exp(a - b));
    } else {
        return (a == -INFINITY) ? b : a + log1p(exp(b - a));
    }
}

/* Ln::map_sum — src/math/mod.rs:62-76 */
double orc_ln_sum(const double* v, size_t n) {
    if (n == 0) return -INFINITY;
    if (n == 1) return v[0];
    double m = -INFINITY;
    for (size_t i = 0; i < n; i++) m = fmax(m, v[i]);
    if (isinf(m)) return m;
    double s = 0.0;
    for (size_t i = 0; i < n; i++) s += exp(v[i] - m);
    return m + log(s);
}

/* Ln::map_sum_init — src/math/mod.rs:80-94 */
double orc_ln_sum_init(const double* v, size_t n, double init) {
    if (n == 0) return init;
    if (n == 1) return orc_ln_add(init, v[0]);
    double m = init;
    for (size_t i = 0; i < n; i++) m = fmax(m, v[i]);
    if (isinf(m)) return m;
    double s = exp(init - m);
    for (size_t i = 0; i < n; i++) s += exp(v[i] - m);
    return m + log(s);
}

/* NBinom::new — src/math/distr/nbinom.rs:35-42 */
orc_nbinom orc_nbinom_new(double n, double p) {
    orc_nbinom d;
    d.n = n; d.p = p;
    d.lnq = log1p(-p);
    d.lnpmf_const = n * log(p) - orc_ln_gamma(n);
    return d;
}

/* NBinom::ln_pmf — nbinom.rs:128-131 */
double orc_nbinom_ln_pmf(const orc_nbinom* d, uint32_t k) {
    double x = (double)k;
    return d->lnpmf_const + orc_ln_gamma(d->n + x) - orc_ln_gamma(x + 1.0) + x * d->lnq;
}

/* NBinom::mode — nbinom.rs:78-80 */
uint32_t orc_nbinom_mode(const orc_nbinom* d) {
    double v = floor((d->n - 1.0) * (1.0 - d->p) / d->p);
    if (!(v > 0.0)) v = 0.0;
    return (uint32_t)v;
}

/* NBinom::cdf — nbinom.rs:145-147 */
double orc_nbinom_cdf(const orc_nbinom* d, uint32_t k) {
    return orc_beta_reg(d->n, (double)(k + 1), d->p);
}

/* WithQuantile::quantile — src/math/distr/mod.rs:38-75 */
double orc_nbinom_quantile(const orc_nbinom* d, double q) {
    if (q <= 0.0) return 0.0;
    if (q >= 1.0) return INFINITY;
    double mean = d->n * (1.0 - d->p) / d->p;
    int64_t low = 0;
    int64_t high = (int64_t)(uint32_t)(2.0 * mean);
    while (orc_nbinom_cdf(d, (uint32_t)high) < q) { low = high; high = high ? high * 2 : 1; }
    while (high >= low) {
        int64_t mid = (low + high) / 2;
        if (orc_nbinom_cdf(d, (uint32_t)mid) >= q) high = mid - 1; else low = mid + 1;
    }
    if (high < 0) return 0.0;   /* the reference would underflow u32 here; unreachable for q > pmf(0) */
    double cdf0 = orc_nbinom_cdf(d, (uint32_t)high);
    double cdf1 = orc_nbinom_cdf(d, (uint32_t)high + 1);
    double diff = cdf1 - cdf0;
    double x = (double)high;
    if (diff == 0.0) return x;
    double r = (q - cdf0) / diff;
    return x * (1.0 - r) + (x + 1.0) * r;
}

/* cache_size — src/bg/insertsz.rs:39-42. Not result-bearing: LinearCache::ln_pmf returns
 * inner.ln_pmf(k) on either side of the boundary (math/distr/lincache.rs:41-48). */
size_t orc_insert_cache_size(const orc_nbinom* d) {
    double q = orc_nbinom_quantile(d, 0.99999);
    size_t v = (q >= 65536.0) ? 65536 : (size_t)q;
    return MIN((size_t)65536, v);
}

/* BetaBinomial — src/math/distr/betabinom.rs:20-102 */
static double bb_ln_pmf_inner(double alpha, double beta, double k, double n) {
    return -orc_ln_beta(n - k + 1.0, k + 1.0) + orc_ln_beta(k + alpha, n - k + beta);
}

void orc_betabinom_inv_cdf2(double alpha, double beta, uint32_t n, double cdf1, double cdf2,
                            uint32_t* out1, uint32_t* out2) {
    double m = (double)n;
    double ln_beta_ab = orc_ln_beta(alpha, beta);
    double const_term = -log(m + 1.0) - ln_beta_ab;
    double ln_cdf = -orc_ln_beta(m + 1.0, 1.0) + orc_ln_beta(alpha, m + beta) + const_term;
    uint32_t k1 = n;
    for (uint32_t i = 0; i < n; i++) {
        double k = (double)(i + 1);
        ln_cdf = orc_ln_add(ln_cdf, bb_ln_pmf_inner(alpha, beta, k, m) + const_term);
        if (exp(ln_cdf) > cdf1) { k1 = i; break; }
    }
    if (exp(ln_cdf) > cdf2) { *out1 = k1; *out2 = k1; return; }
    for (uint32_t i = k1 + 1; i < n; i++) {
        double k = (double)(i + 1);
        ln_cdf = orc_ln_add(ln_cdf, bb_ln_pmf_inner(alpha, beta, k, m) + const_term);
        if (exp(ln_cdf) > cdf2) { *out1 = k1; *out2 = i; return; }
    }
    *out1 = k1; *out2 = n;
}

/* EditDistCache::get_anew — src/bg/err_prof.rs:434-443 */
void orc_edit_thresholds(const lcty_bg* bg, uint32_t read_len, uint32_t* good, uint32_t* passable) {
    if (bg->edit_kind == LCTY_EDIT_FRACTION) {
        double rl = (double)read_len;
        *good = (uint32_t)(rl * bg->edit_p1);
        *passable = (uint32_t)(rl * bg->edit_p2);
    } else {
        orc_betabinom_inv_cdf2(bg->edit_alpha, bg->edit_beta, read_len, bg->edit_p1, bg->edit_p2, good, passable);
    }
}

/* DistrCache::new + BayesCalc::ln_pmf — src/model/distr_cache.rs:61-75, src/math/distr/bayes.rs:27-35 */
double orc_depth_ln_pmf(const lcty_bg* bg, const lcty_params* prm, uint32_t gc, uint32_t depth) {
    double mul_coef = bg->is_paired ? 2.0 : 1.0;
    orc_nbinom cn1 = orc_nbinom_new(bg->depth_n[gc] * mul_coef, bg->depth_p[gc]);   /* NBinom::mul nbinom.rs:68-70 */
    double null_prob = orc_nbinom_ln_pmf(&cn1, depth);
    double probs[LCTY_MAX_ALT_CN + 1];
    for (uint32_t i = 0; i < prm->n_alt_cn; i++) {
        orc_nbinom alt = orc_nbinom_new(cn1.n * prm->alt_cn[i], cn1.p);
        probs[i] = orc_nbinom_ln_pmf(&alt, depth);
    }
    double sum_prob = orc_ln_sum_init(probs, prm->n_alt_cn, null_prob);
    return null_prob - sum_prob;
}

/* statrs StudentsT::cdf (location 0, scale 1) */
double orc_students_t_cdf(double freedom, double x) {
    if (isinf(freedom)) return 0.5 * erfc(-x / sqrt(2.0));
    double h = freedom / (freedom + x * x);
    double ib = 0.5 * orc_beta_reg(freedom / 2.0, 0.5, h);
    return x <= 0.0 ? ib : 1.0 - ib;
}

/* unpaired_onesided_t_test::<false> — src/math/mod.rs:180-198 */
double orc_t_test(double mean1, double var1, double mean2, double var2, double n) {
    double var_sum = var1 + var2;
    double t_stat = (mean1 - mean2) * sqrt(n / var_sum);
    double freedom = (n - 1.0) * var_sum * var_sum / (var1 * var1 + var2 * var2);
    return orc_students_t_cdf(freedom, t_stat);
}

/* unpaired_onesided_t_test_diffsizes::<false> — src/math/mod.rs:200-220 */
double orc_t_test_diffsizes(double mean1, double var1, double mean2, double var2, double n1, double n2) {
    double nvar1 = var1 / n1, nvar2 = var2 / n2;
    double sum_nvar = nvar1 + nvar2;
    double t_stat = (mean1 - mean2) / sqrt(sum_nvar);
    double freedom = sum_nvar * sum_nvar / (nvar1 * nvar1 / (n1 - 1.0) + nvar2 * nvar2 / (n2 - 1.0));
    return orc_students_t_cdf(freedom, t_stat);
}

/* ======================================================================== */
/* params                                                                    */
/* ======================================================================== */

static const double LN10 = 2.302585092994045684;

/* model::Params::default — src/model/mod.rs:108-135 */
void orc_params_default(lcty_params* p) {
    memset(p, 0, sizeof(*p));
    p->boundary_size = 200;
    p->tweak = -1;
    p->lik_skew = 0.85;
    p->prob_diff = NAN;
    p->unmapped_penalty = NAN;
    p->poor_compl = 0.5;
    p->poor_compl_edit = 0.7;
    p->compl_weight_bp = 0.5; p->compl_weight_pow = 4.0;
    p->kmers_weight_bp = 0.2; p->kmers_weight_pow = 4.0;
    p->min_weight = 0.001;
    p->filt_diff = 100.0 * LN10;
    p->prob_thresh = -4.0 * LN10;
    p->alt_cn[0] = 0.3; p->alt_cn[1] = 2.0; p->alt_cn[2] = 3.0; p->alt_cn[3] = 4.0; p->alt_cn[4] = 5.0;
    p->n_alt_cn = 5;
    p->kmer_soft_thresh = 5;
    p->kmer_hard_thresh = 1;
    p->complexity_k = 5;
    p->threads = 8;    /* src/command/genotype.rs:127 */
}

/* set_tweak_size (model/mod.rs:179-197) + genotype.rs:1291-1296 */
int orc_params_resolve(lcty_params* p, const lcty_bg* bg) {
    if (p->tweak < 0) {
        uint32_t t = (uint32_t)round((double)bg->window * 0.5);
        t = MIN(t, 200u);
        t = MIN(t, p->boundary_size ? p->boundary_size - 1 : 0);
        p->tweak = (int32_t)t;
    }
    if ((uint32_t)p->tweak >= p->boundary_size) return LCTY_ERR_INVALID_INPUT;
    if ((uint32_t)p->tweak > 65535u / 2 - 1) return LCTY_ERR_INVALID_INPUT;
    if (isnan(p->unmapped_penalty))
        p->unmapped_penalty = (bg->technology == LCTY_TECH_ILLUMINA ? -10.0 : -100.0) * LN10;
    if (isnan(p->prob_diff))
        p->prob_diff = fabs(p->unmapped_penalty) + 1.0 * LN10;
    p->prob_diff = fabs(p->prob_diff);   /* Params::validate model/mod.rs:148 */
    return LCTY_OK;
}

/* ======================================================================== */
/* k-mers                                                                    */
/* ======================================================================== */

/* kmers::kmers — src/seq/kmers.rs:163-202 */
#define DEFINE_KMERS(NAME, T)                                                                \
size_t NAME(const uint8_t* seq, size_t n, uint32_t k, int canonical, T* out) {               \
    T mask = (T)(((T)1 << (2 * k)) - (T)1);                                                   \
    uint32_t rv_shift = canonical ? 2 * k - 2 : 0;                                            \
    T fw = 0, rv = 0;                                                                         \
    uint32_t k_1 = k - 1;                                                                     \
    uint64_t reset = k_1;                                                                     \
    size_t w = 0;                                                                             \
    for (size_t idx = 0; idx < n; idx++) {                                                    \
        uint64_t i = idx;                                                                     \
        uint8_t nt = seq[idx];                                                                \
        uint8_t enc;                                                                          \
        switch (nt) {                                                                         \
            case 'A': enc = 0; break;                                                         \
            case 'C': enc = 1; break;                                                         \
            case 'G': enc = 2; break;                                                         \
            case 'T': enc = 3; break;                                                         \
            default:                                                                          \
                reset = i + k;                                                                \
                if (i + 1 >= k) out[w++] = (T)~(T)0;                                          \
                continue;                                                                     \
        }                                                                                     \
        fw = (T)(((T)(fw << 2) | (T)enc) & mask);                                             \
        if (canonical) rv = (T)((rv >> 2) | ((T)(3 - enc) << rv_shift));                      \
        if (i >= reset) {                                                                     \
            out[w++] = (canonical && rv < fw) ? rv : fw;                                      \
        } else if (i + 1 >= k) {                                                              \
            out[w++] = (T)~(T)0;                                                              \
        }                                                                                     \
    }                                                                                         \
    return w;                                                                                 \
}
DEFINE_KMERS(orc_kmers_u128, orc_u128)
DEFINE_KMERS(orc_kmers_u32, uint32_t)

/* linguistic_complexity — src/seq/compl.rs:115-140 (+ remove_add_kmer 36-48).
 * Returns the integer `unique` per window; the reference stores unique*mult. */
size_t orc_complexity_counts(const uint8_t* seq, size_t n, uint32_t k, uint32_t w, uint16_t* out) {
    size_t nk = n + 1 - k;
    uint32_t* kmers = (uint32_t*)malloc(sizeof(uint32_t) * nk);
    orc_kmers_u32(seq, n, k, 0, kmers);
    /* IntMap<u32,u16>: k-mers < 4^k plus the UNDEF value -> dense table + one extra slot */
    size_t tbl = ((size_t)1 << (2 * k)) + 1;
    uint16_t* counts = (uint16_t*)calloc(tbl, sizeof(uint16_t));
#define SLOT(x) ((x) == 0xFFFFFFFFu ? tbl - 1 : (size_t)(x))
    uint16_t unique = 0;
    size_t first = w - k + 1;
    for (size_t i = 0; i < first; i++) {
        uint16_t* c = &counts[SLOT(kmers[i])];
        unique += (*c == 0);
        (*c)++;
    }
    size_t o = 0;
    out[o++] = unique;
    for (size_t i = 0; i + first < nk; i++) {
        uint32_t rem = kmers[i], add = kmers[i + first];
        if (rem != add) {
            uint16_t* c1 = &counts[SLOT(add)];
            unique += (*c1 == 0);
            (*c1)++;
            uint16_t* c2 = &counts[SLOT(rem)];
            unique -= (*c2 == 1);
            (*c2)--;
        }
        out[o++] = unique;
    }
#undef SLOT
    free(counts);
    free(kmers);
    return o;
}

/* ---- HashSet<u128> for UniqueKmers (locs.rs:919): open addressing ---------- */

static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
static inline size_t u128_hash(orc_u128 k) {
    return (size_t)mix64((uint64_t)k ^ mix64((uint64_t)(k >> 64) + 0x9e3779b97f4a7c15ULL));
}
static void set_init(u128set* s, size_t cap_pow2) {
    s->cap = cap_pow2; s->len = 0;
    s->keys = (orc_u128*)malloc(sizeof(orc_u128) * s->cap);
    s->used = (uint8_t*)calloc(s->cap, 1);
}
static void set_free(u128set* s) { free(s->keys); free(s->used); }
static int set_contains(const u128set* s, orc_u128 k) {
    size_t i = u128_hash(k) & (s->cap - 1);
    while (s->used[i]) {
        if (s->keys[i] == k) return 1;
        i = (i + 1) & (s->cap - 1);
    }
    return 0;
}
static void set_insert_nogrow(u128set* s, orc_u128 k) {
    size_t i = u128_hash(k) & (s->cap - 1);
    while (s->used[i]) {
        if (s->keys[i] == k) return;
        i = (i + 1) & (s->cap - 1);
    }
    s->used[i] = 1; s->keys[i] = k; s->len++;
}
static void set_insert(u128set* s, orc_u128 k) {
    if ((s->len + 1) * 2 > s->cap) {
        u128set n; set_init(&n, s->cap * 2);
        for (size_t i = 0; i < s->cap; i++) if (s->used[i]) set_insert_nogrow(&n, s->keys[i]);
        set_free(s); *s = n;
    }
    set_insert_nogrow(s, k);
}

/* ======================================================================== */
/* locus: ContigSet + ContigInfos + UniqueKmers + distributions              */
/* ======================================================================== */



const lcty_params* orc_locus_params(const orc_locus* l) { return &l->prm; }

/* ExplicitWeights — model/windows.rs:196-250. SCALE = 2^32; the running sums are integers, so sums over ranges are exact. */
#define EW_SCALE 4294967296.0
double orc_explicit_average(const orc_locus* l, uint32_t allele, uint32_t i, uint32_t j) {     /* windows.rs:236-238 */
    const uint64_t* cum = l->ew_cum[allele];
    return (double)((cum[j] - cum[i]) / (uint64_t)(j - i)) / EW_SCALE;
}
/* ContigInfo::read_end_weight — windows.rs:493-503: the largest of the values at the middle of the read end and half a window
 * to either side (weights.len() counts the entry finish() appends) */
double orc_read_end_weight(const orc_locus* l, uint32_t allele, uint32_t middle) {
    if (middle == LCTY_NONE_U32) return 0.0;
    const double* val = l->ew_val[allele];
    const uint32_t n = l->infos[allele].len + 1;
    const uint32_t u = l->bg.window / 2;
    const uint32_t lo = middle > u ? middle - u : 0;                       /* saturating_sub */
    const uint32_t hi = middle + u < n - 1 ? middle + u : n - 1;
    return fmax(fmax(val[middle], val[lo]), val[hi]);
}

static void free_explicit(orc_locus* l) {
    if (l->ew_val) for (uint32_t a = 0; a < l->n_alleles; a++) free(l->ew_val[a]);
    if (l->ew_cum) for (uint32_t a = 0; a < l->n_alleles; a++) free(l->ew_cum[a]);
    free(l->ew_val); free(l->ew_cum);
    l->ew_val = NULL; l->ew_cum = NULL; l->has_explicit = 0;
}

/* load_explicit_weights — windows.rs:257-317 (lines already split into fields) */
int orc_locus_set_explicit_weights(orc_locus* l, uint32_t n, const uint32_t* allele, const uint32_t* start, const uint32_t* end,
                                   const double* value) {
    const uint32_t A = l->n_alleles;
    free_explicit(l);
    l->ew_val = (double**)calloc(A, sizeof(double*));
    l->ew_cum = (uint64_t**)calloc(A, sizeof(uint64_t*));
    uint32_t* filled = (uint32_t*)calloc(A, sizeof(uint32_t));
    uint64_t* sum = (uint64_t*)calloc(A, sizeof(uint64_t));
    int err = 0;
    for (uint32_t a = 0; a < A; a++) {
        l->ew_val[a] = (double*)malloc(sizeof(double) * ((size_t)l->infos[a].len + 1));
        l->ew_cum[a] = (uint64_t*)malloc(sizeof(uint64_t) * ((size_t)l->infos[a].len + 1));
    }
    for (uint32_t t = 0; t < n && !err; t++) {
        const uint32_t a = allele[t];
        if (a >= A) continue;                                                /* unknown contig: line ignored (269-272) */
        if (start[t] >= end[t] || end[t] > l->infos[a].len) { err = 2; break; }   /* InvalidInput, interv.rs:112-116 (an empty interval: Interval::new asserts) */
        if (!(value[t] >= 0.0 && value[t] <= 1.0)) { err = 1; break; }       /* 285-288 */
        if (filled[a] != start[t]) { err = 1; break; }                       /* "not fully covered", 291-295 */
        const uint64_t inc = (uint64_t)(value[t] * EW_SCALE);                /* extend_by, 212-218 */
        for (uint32_t i = start[t]; i < end[t]; i++) {
            l->ew_val[a][i] = value[t];
            l->ew_cum[a][i] = sum[a];
            sum[a] += inc;
        }
        filled[a] = end[t];
    }
    for (uint32_t a = 0; a < A && !err; a++) {
        if (filled[a] == 0 || filled[a] != l->infos[a].len) { err = 1; break; }   /* missing / different length, 305-313 */
        l->ew_val[a][filled[a]] = l->ew_val[a][filled[a] - 1];              /* finish(), 221-224 */
        l->ew_cum[a][filled[a]] = sum[a];
    }
    free(filled); free(sum);
    if (err) { free_explicit(l); return err == 2 ? LCTY_ERR_INVALID_INPUT : LCTY_ERR_INVALID_DATA; }
    l->has_explicit = 1;
    return LCTY_OK;
}

/* ContigInfo::new — src/model/windows.rs:362-424 (explicit weights: orc_locus_set_explicit_weights) */
static int contig_info_new(orc_locus* l, uint32_t a, const uint16_t* counts, size_t n_counts) {
    orc_contig_info* ci = &l->infos[a];
    const uint8_t* seq = l->seqs[a];
    uint32_t contig_len = (uint32_t)(l->seq_off[a + 1] - l->seq_off[a]);
    uint32_t window = l->bg.window, neighb = l->bg.neighb;
    if (contig_len < window + 2 * l->prm.boundary_size) return LCTY_ERR_RUNTIME;   /* windows.rs:375-378 */
    if (neighb > contig_len || neighb + 1 < l->k + 1) return LCTY_ERR_RUNTIME;
    ci->len = contig_len;
    ci->n_windows = (contig_len - 2 * l->prm.boundary_size) / window;
    uint32_t sum_len = ci->n_windows * window;
    ci->reg_start = (contig_len - sum_len) / 2;
    ci->reg_end = ci->reg_start + sum_len;
    ci->n_pos = contig_len - neighb + 1;
    ci->gc = (uint8_t*)malloc(ci->n_pos);
    ci->uniq_cnt = (uint32_t*)malloc(sizeof(uint32_t) * ci->n_pos);
    ci->compl_cnt = (uint16_t*)malloc(sizeof(uint16_t) * ci->n_pos);

    /* cumul_sums (ext/vec.rs:236-246): c[0] = 0, c[i+1] = c[i] + x[i] */
    uint32_t* cum = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)contig_len + 1));
    cum[0] = 0;
    for (uint32_t i = 0; i < contig_len; i++) cum[i + 1] = cum[i] + (seq[i] == 'C' || seq[i] == 'G');
    double mult = 100.0 / (double)neighb;
    for (uint32_t i = 0; i < ci->n_pos; i++)
        ci->gc[i] = (uint8_t)round(mult * (double)(cum[i + neighb] - cum[i]));     /* windows.rs:388-391 */

    if (n_counts != (size_t)contig_len + 1 - l->k) { free(cum); return LCTY_ERR_INVALID_DATA; } /* locs.rs:944 */
    cum[0] = 0;
    for (size_t i = 0; i < n_counts; i++) cum[i + 1] = cum[i] + (counts[i] == 0);
    uint32_t span = neighb + 1 - l->k;
    for (uint32_t i = 0; i < ci->n_pos; i++) ci->uniq_cnt[i] = cum[i + span] - cum[i];           /* windows.rs:395-403 */
    free(cum);

    size_t nc = orc_complexity_counts(seq, contig_len, l->prm.complexity_k, neighb, ci->compl_cnt); /* windows.rs:404 */
    if (nc != ci->n_pos) return LCTY_ERR_RUNTIME;
    return LCTY_OK;
}

orc_locus* orc_locus_new(uint32_t n_alleles, const uint8_t* seqs, const uint64_t* seq_off,
                         const uint16_t* offtarget, const uint64_t* cnt_off, uint32_t k,
                         const lcty_bg* bg, const lcty_params* params) {
    orc_locus* l = (orc_locus*)calloc(1, sizeof(orc_locus));
    l->n_alleles = n_alleles; l->k = k; l->bg = *bg; l->prm = *params;
    uint64_t total = seq_off[n_alleles];
    l->seq_store = (uint8_t*)malloc(total ? total : 1);
    memcpy(l->seq_store, seqs, total);
    l->seq_off = (uint64_t*)malloc(sizeof(uint64_t) * (n_alleles + 1));
    memcpy(l->seq_off, seq_off, sizeof(uint64_t) * (n_alleles + 1));
    l->seqs = (const uint8_t**)malloc(sizeof(uint8_t*) * n_alleles);
    for (uint32_t a = 0; a < n_alleles; a++) l->seqs[a] = l->seq_store + seq_off[a];

    l->left_padding = (bg->neighb - bg->window) / 2;          /* windows.rs:384 */
    l->half_neighb = bg->neighb / 2;                           /* windows.rs:422 */
    l->uniq_mult = 1.0 / (double)(bg->neighb + 1 - k);         /* windows.rs:397 */
    {
        size_t ck = params->complexity_k;
        size_t a1 = (size_t)bg->neighb + 1 - ck, a2 = (size_t)1 << (2 * ck);
        l->compl_mult = 1.0 / (double)MIN(a1, a2);             /* compl.rs:124 */
    }

    /* UniqueKmers::new — locs.rs:930-963 */
    set_init(&l->unique, 1 << 16);
    l->infos = (orc_contig_info*)calloc(n_alleles, sizeof(orc_contig_info));
    size_t max_len = 0;
    for (uint32_t a = 0; a < n_alleles; a++) max_len = MAX(max_len, (size_t)(seq_off[a + 1] - seq_off[a]));
    orc_u128* buf = (orc_u128*)malloc(sizeof(orc_u128) * (max_len + 1));
    int err = 0;
    for (uint32_t a = 0; a < n_alleles && !err; a++) {
        size_t len = (size_t)(seq_off[a + 1] - seq_off[a]);
        size_t nk = orc_kmers_u128(l->seqs[a], len, k, 1, buf);
        const uint16_t* cnt = offtarget + cnt_off[a];
        size_t n_counts = (size_t)(cnt_off[a + 1] - cnt_off[a]);
        if (nk != n_counts) { err = 1; break; }
        for (size_t i = 0; i < nk; i++) if (cnt[i] == 0) set_insert(&l->unique, buf[i]);
        if (contig_info_new(l, a, cnt, n_counts) != LCTY_OK) err = 1;
    }
    free(buf);
    if (err) { orc_locus_free(l); return NULL; }
    l->weight_mult = 1.0 / (double)(params->kmer_soft_thresh + 1 - params->kmer_hard_thresh);
    l->weight_interc = (1.0 - (double)params->kmer_hard_thresh) * l->weight_mult;

    /* InsertDistr::load — bg/insertsz.rs:195-208 */
    if (bg->is_paired) {
        l->ins = orc_nbinom_new(bg->ins_n, bg->ins_p);
        l->ins_lut_size = orc_insert_cache_size(&l->ins);
        l->ins_lut = (double*)malloc(sizeof(double) * (l->ins_lut_size ? l->ins_lut_size : 1));
        for (size_t i = 0; i < l->ins_lut_size; i++) l->ins_lut[i] = orc_nbinom_ln_pmf(&l->ins, (uint32_t)i);
        l->ins_mode_prob = orc_nbinom_ln_pmf(&l->ins, orc_nbinom_mode(&l->ins));
    } else {
        l->ins_mode_prob = NAN;
    }
    l->depth_lut = (double*)malloc(sizeof(double) * LCTY_GC_BINS * LCTY_DEPTH_CACHE);
    for (uint32_t gc = 0; gc < LCTY_GC_BINS; gc++)
        for (uint32_t d = 0; d < LCTY_DEPTH_CACHE; d++)
            l->depth_lut[gc * LCTY_DEPTH_CACHE + d] = orc_depth_ln_pmf(&l->bg, &l->prm, gc, d);
    return l;
}

void orc_locus_free(orc_locus* l) {
    if (!l) return;
    if (l->infos) for (uint32_t a = 0; a < l->n_alleles; a++) {
        free(l->infos[a].gc); free(l->infos[a].uniq_cnt); free(l->infos[a].compl_cnt);
    }
    free_explicit(l);
    free(l->depth_lut); free(l->depth_ext_inj); free(l->win_weight_inj); free(l->ci_off_inj); free(l->infos); free(l->seqs); free(l->seq_store); free(l->seq_off); free(l->ins_lut);
    set_free(&l->unique);
    free(l);
}

uint64_t orc_locus_n_unique_kmers(const orc_locus* l) {
    /* the UNDEF k-mer can enter the set only through a window with N whose count is 0 */
    return l->unique.len;
}

int orc_locus_contig_info(const orc_locus* l, uint32_t a, uint8_t* gc, uint32_t* uniq_cnt,
                          uint16_t* compl_cnt, uint32_t* n_windows, uint32_t* reg_start) {
    if (a >= l->n_alleles) return LCTY_ERR_INVALID_INPUT;
    const orc_contig_info* ci = &l->infos[a];
    if (gc) memcpy(gc, ci->gc, ci->n_pos);
    if (uniq_cnt) memcpy(uniq_cnt, ci->uniq_cnt, sizeof(uint32_t) * ci->n_pos);
    if (compl_cnt) memcpy(compl_cnt, ci->compl_cnt, sizeof(uint16_t) * ci->n_pos);
    if (n_windows) *n_windows = ci->n_windows;
    if (reg_start) *reg_start = ci->reg_start;
    return LCTY_OK;
}

/* InsertDistr::ln_prob -> LinearCache::ln_pmf — insertsz.rs:153-155, lincache.rs:41-48 */
double orc_locus_insert_lnprob(const orc_locus* l, uint32_t sz) {
    if ((size_t)sz < l->ins_lut_size) return l->ins_lut[sz];
    return orc_nbinom_ln_pmf(&l->ins, sz);
}
double orc_locus_insert_penalty(const orc_locus* l) { return l->ins_mode_prob; }

/* ContigInfo::neighb_complexity — windows.rs:447-452 */
static double neighb_complexity(const orc_locus* l, uint32_t contig, uint32_t middle) {
    const orc_contig_info* ci = &l->infos[contig];
    uint32_t s = middle > l->half_neighb ? middle - l->half_neighb : 0;
    uint32_t i = MIN(s, ci->n_pos - 1);
    return (double)ci->compl_cnt[i] * l->compl_mult;
}

/* ======================================================================== */
/* AllAlignments::load                                                       */
/* ======================================================================== */

typedef struct {
    uint32_t start, end;     /* Interval */
    uint32_t rec_ix;         /* index of the record inside the pair (input order); 0x80000000 | k for the k-th transferred alignment */
    uint16_t contig;
    uint8_t read_end;        /* 0 / 1 */
    uint8_t reverse;
    uint32_t edit, read_len; /* EditDist */
    double ln_prob;
    uint32_t* own_cigar;     /* raw CIGAR words of a transferred alignment (NULL: the input record rec_ix) */
    uint32_t own_n;
} o_aln;

typedef struct { uint64_t key; uint32_t index, pos; } pos_entry;   /* PosCollection, locs.rs:186-217 */

typedef struct {
    o_aln* alns; size_t n_alns, cap_alns;
    pos_entry* pos; size_t n_pos, cap_pos;
    /* open-addressing index over pos[] standing in for IntMap<u64, PosCollectionValue> */
    uint32_t* hslot; uint32_t* hgen; size_t hcap; uint32_t gen;
    uint32_t good_dist[2], passable_dist[2], best_edit[2];
    double best_lik[2];
    uint32_t** owned; size_t n_owned, cap_owned;       /* CIGARs of transferred alignments, freed with the read */
} prelim;

#define NOT_SAVED 0xFFFFFFFFu

static void prelim_reset(prelim* p, size_t n_records) {
    for (size_t i = 0; i < p->n_owned; i++) free(p->owned[i]);
    p->n_owned = 0;
    p->n_alns = 0; p->n_pos = 0;
    size_t need = 16;
    while (need < 2 * n_records + 2) need <<= 1;
    if (need > p->hcap) {
        free(p->hslot); free(p->hgen);
        p->hcap = need;
        p->hslot = (uint32_t*)malloc(sizeof(uint32_t) * need);
        p->hgen = (uint32_t*)calloc(need, sizeof(uint32_t));
        p->gen = 0;
    }
    p->gen++;
    for (int e = 0; e < 2; e++) {
        p->good_dist[e] = 0xFFFFFFFFu; p->passable_dist[e] = 0xFFFFFFFFu; p->best_edit[e] = 0xFFFFFFFFu;
        p->best_lik[e] = -INFINITY;
    }
}


typedef struct {
    const orc_locus* l;
    const lcty_reads_host* in;
    int err;
} load_ctx;

/* Cigar::from_raw + ref_len + op counts; returns 0 on unsupported op.
 * count_region_operations_fast — src/seq/aln.rs:301-317; limited_clipping 288-296;
 * soft_clipping — src/seq/cigar.rs:519-527; hard_to_soft 309-320. */
static int score_record(load_ctx* c, const lcty_aln_rec* rec, const uint32_t* cig, int is_primary,
                        uint8_t read_end, o_aln* out, int* empty) {
    const orc_locus* l = c->l;
    uint32_t n = rec->n_cigar;
    *empty = 0;
    if (n == 0) { *empty = 1; return 1; }
    uint32_t op_first = cig[0] & 15u, op_last = cig[n - 1] & 15u;
    if (is_primary && (op_first == LCTY_CIGAR_H || op_last == LCTY_CIGAR_H)) {
        c->err = LCTY_ERR_INVALID_DATA;    /* assert!(!cigar.has_hard_clipping()) locs.rs:526 */
        return 0;
    }
    uint32_t matches = 0, mism = 0, ins = 0, del = 0, ref_len = 0;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t op = cig[i] & 15u, len = cig[i] >> 4;
        /* hard_to_soft: only the first / last tuple is converted */
        if (op == LCTY_CIGAR_H && (i == 0 || i == n - 1)) op = LCTY_CIGAR_S;
        switch (op) {
            case LCTY_CIGAR_EQ: matches += len; ref_len += len; break;
            case LCTY_CIGAR_X: mism += len; ref_len += len; break;
            case LCTY_CIGAR_D: del += len; ref_len += len; break;
            case LCTY_CIGAR_I: ins += len; break;
            case LCTY_CIGAR_S: break;
            default:
                c->err = LCTY_ERR_INVALID_DATA;   /* panic!("Unsupported CIGAR operation") aln.rs:311 */
                return 0;
        }
    }
    uint32_t left = 0, right = 0;
    {
        uint32_t f = op_first == LCTY_CIGAR_H ? LCTY_CIGAR_S : op_first;
        uint32_t la = op_last == LCTY_CIGAR_H ? LCTY_CIGAR_S : op_last;
        if (f == LCTY_CIGAR_S) left = cig[0] >> 4;
        if (la == LCTY_CIGAR_S) right = cig[n - 1] >> 4;
    }
    if (rec->contig >= l->n_alleles) { c->err = LCTY_ERR_INVALID_DATA; return 0; }
    uint32_t contig_len = l->infos[rec->contig].len;
    uint32_t start = rec->pos, end = rec->pos + ref_len;
    uint32_t clip = MIN(left, start) + MIN(right, contig_len > end ? contig_len - end : 0);
    /* OperCounts::edit_distance — bg/err_prof.rs:73-79 */
    uint32_t common = mism + ins + clip;
    out->edit = common + del;
    out->read_len = common + matches;
    /* ErrorProfile::ln_prob — bg/err_prof.rs:212-221 (fixed evaluation order) */
    const double* lp = l->bg.op_lnprobs;
    out->ln_prob = lp[0] * (double)matches + lp[1] * (double)mism + lp[2] * (double)ins
                   + lp[3] * (double)del + lp[4] * (double)clip;
    out->start = start; out->end = end;
    out->own_cigar = NULL; out->own_n = 0;
    out->contig = rec->contig;
    out->read_end = read_end;
    out->reverse = (rec->flags & LCTY_FLAG_REVERSE) != 0;
    return 1;
}

/* PrelimAlignments::push — src/model/locs.rs:298-344 */
static int prelim_push(prelim* p, const o_aln* aln) {
    int e = aln->read_end;
    p->best_edit[e] = MIN(p->best_edit[e], aln->edit);
    p->best_lik[e] = fmax(p->best_lik[e], aln->ln_prob);
    uint32_t new_ix = (uint32_t)p->n_alns;
    int save = aln->edit <= p->passable_dist[e];
    if (new_ix == 0 && !save) return 0;

    /* encode — locs.rs:181-184 */
    uint64_t key = ((uint64_t)(e + 1) << 48) | ((uint64_t)aln->contig << 32) | (uint64_t)(aln->start >> 7);
    pos_entry* ent = NULL;
    size_t h = (size_t)mix64(key) & (p->hcap - 1);
    while (p->hgen[h] == p->gen) {
        if (p->pos[p->hslot[h]].key == key) { ent = &p->pos[p->hslot[h]]; break; }
        h = (h + 1) & (p->hcap - 1);
    }
#define PUSH_ALN() do {                                                             \
        if (p->n_alns == p->cap_alns) {                                             \
            p->cap_alns = p->cap_alns ? p->cap_alns * 2 : 64;                       \
            p->alns = (o_aln*)realloc(p->alns, sizeof(o_aln) * p->cap_alns);        \
        }                                                                           \
        p->alns[p->n_alns++] = *aln;                                                \
    } while (0)
    if (ent) {
        if (save) {
            if (ent->index == NOT_SAVED) {
                ent->index = new_ix; ent->pos = aln->start;
                PUSH_ALN();
            } else if (aln->ln_prob > p->alns[ent->index].ln_prob) {
                p->alns[ent->index] = *aln;
                ent->pos = aln->start;
            }
        }
    } else {
        if (p->n_pos == p->cap_pos) {
            p->cap_pos = p->cap_pos ? p->cap_pos * 2 : 64;
            p->pos = (pos_entry*)realloc(p->pos, sizeof(pos_entry) * p->cap_pos);
        }
        p->hgen[h] = p->gen; p->hslot[h] = (uint32_t)p->n_pos;
        pos_entry* ne = &p->pos[p->n_pos++];
        ne->key = key; ne->pos = aln->start;
        if (save) { ne->index = new_ix; PUSH_ALN(); } else { ne->index = NOT_SAVED; }
    }
#undef PUSH_ALN
    return save;
}

/* read_next_alns — src/model/locs.rs:502-567. `ri` is advanced past this end's records. */
static int read_next_alns(load_ctx* c, uint64_t pair, uint64_t* ri, uint64_t r_end, uint8_t read_end,
                          double* weight, prelim* p) {
    const orc_locus* l = c->l;
    const lcty_reads_host* in = c->in;
    const uint32_t* cig_base = in->cigar + in->cigar_off[pair];
    uint64_t first = in->aln_off[pair];
    if (*ri >= r_end) { c->err = LCTY_ERR_INVALID_DATA; return 0; }   /* expect("Cannot read any more records") */
    const lcty_aln_rec* rec = &in->recs[*ri];
    uint32_t read_len = in->mate_len[2 * pair + read_end];
    if (read_len == 0) { c->err = LCTY_ERR_INVALID_DATA; return 0; }  /* locs.rs:511-517 */
    if (rec->flags & (LCTY_FLAG_SECONDARY | LCTY_FLAG_SUPPL)) { c->err = LCTY_ERR_INVALID_DATA; return 0; }
#define SKIP_UNTIL_PRIMARY() do { (*ri)++;                                                       \
        while (*ri < r_end && (in->recs[*ri].flags & (LCTY_FLAG_SECONDARY | LCTY_FLAG_SUPPL))) (*ri)++; \
    } while (0)
    if (rec->flags & LCTY_FLAG_UNMAPPED) {       /* locs.rs:520-523 */
        (*ri)++;
        return 0;
    }
    o_aln aln; int empty;
    if (!score_record(c, rec, cig_base + rec->cigar_rel, 1, read_end, &aln, &empty)) return 0;
    if (empty) { c->err = LCTY_ERR_INVALID_DATA; return 0; }          /* tuples[0] would panic, locs.rs:526 */
    aln.rec_ix = (uint32_t)(*ri - first);
    /* locs.rs:529-536 */
    double compl = l->bg.technology == LCTY_TECH_ILLUMINA
        ? neighb_complexity(l, aln.contig, (aln.start + aln.end) / 2) : 1.0;
    uint32_t good, passable;
    orc_edit_thresholds(&l->bg, read_len, &good, &passable);
    uint32_t threshold = good;
    if (compl <= l->prm.poor_compl) {
        threshold = MAX(good, (uint32_t)(l->prm.poor_compl_edit * (double)read_len));
        passable += threshold - good;
    }
    p->good_dist[read_end] = threshold;
    p->passable_dist[read_end] = passable;
    if (!prelim_push(p, &aln)) {                 /* locs.rs:539-543 */
        SKIP_UNTIL_PRIMARY();
        return 0;
    }
    (*ri)++;
    while (*ri < r_end && (in->recs[*ri].flags & (LCTY_FLAG_SECONDARY | LCTY_FLAG_SUPPL))) {   /* locs.rs:545-558 */
        rec = &in->recs[*ri];
        if (!score_record(c, rec, cig_base + rec->cigar_rel, 0, read_end, &aln, &empty)) return 0;
        if (!empty) {
            aln.rec_ix = (uint32_t)(*ri - first);
            prelim_push(p, &aln);
        }
        (*ri)++;
    }
    uint32_t best_edit = p->best_edit[read_end];
    uint32_t req = l->prm.strict_subset ? passable : threshold;      /* locs.rs:560-564 */
    if (best_edit > req) return 0;
    *weight *= best_edit <= good ? 1.0 : sqrt((double)good / (double)best_edit);   /* locs.rs:565 */
    return 1;
#undef SKIP_UNTIL_PRIMARY
}

/* UniqueKmers::calculate_read_weight — src/model/locs.rs:968-1002 */
static uint16_t count_unique_kmers(const orc_locus* l, const uint8_t* seq, uint32_t len, orc_u128* buf) {
    size_t nk = orc_kmers_u128(seq, len, l->k, 1, buf);
    uint16_t count = 0;
    size_t i = 0;
    while (i < nk) {
        orc_u128 km = buf[i++];
        if (set_contains(&l->unique, km)) {
            if (count != 0xFFFF) count++;           /* saturating_add */
            i += (size_t)(l->k - 2) + 1;            /* kmers_iter.nth(k_2) consumes k-1 items */
        }
    }
    return count;
}

static void unpack_mate(const lcty_reads_host* in, uint64_t mate, uint8_t* out) {
    uint64_t off = in->mate_off[mate];
    uint32_t len = in->mate_len[mate];
    for (uint32_t i = 0; i < len; i++) {
        uint64_t b = off + i;
        uint32_t code = (in->bases2[b >> 4] >> (2 * (b & 15))) & 3u;
        int isn = (in->nmask[b >> 5] >> (b & 31)) & 1u;
        out[i] = isn ? 'N' : "ACGT"[code];
    }
}

typedef struct { o_aln aln; } sort_aln;

/* (contig desc, read_end desc, ln_prob asc) so that pop() yields contig asc, end asc, prob desc
 * — locs.rs:819-820. Exact ties are implementation-defined in the reference
 * (sort_unstable); here ties pop in input-record order (rec_ix ascending). */
static int cmp_pe(const void* x, const void* y) {
    const o_aln* a = (const o_aln*)x; const o_aln* b = (const o_aln*)y;
    if (a->contig != b->contig) return a->contig > b->contig ? -1 : 1;
    if (a->read_end != b->read_end) return a->read_end > b->read_end ? -1 : 1;
    if (a->ln_prob != b->ln_prob) return a->ln_prob < b->ln_prob ? -1 : 1;
    if (a->rec_ix != b->rec_ix) return a->rec_ix > b->rec_ix ? -1 : 1;
    return 0;
}
/* single-end: (contig desc, ln_prob asc) — locs.rs:884 */
static int cmp_se(const void* x, const void* y) {
    const o_aln* a = (const o_aln*)x; const o_aln* b = (const o_aln*)y;
    if (a->contig != b->contig) return a->contig > b->contig ? -1 : 1;
    if (a->ln_prob != b->ln_prob) return a->ln_prob < b->ln_prob ? -1 : 1;
    if (a->rec_ix != b->rec_ix) return a->rec_ix > b->rec_ix ? -1 : 1;
    return 0;
}

typedef struct { lcty_pair_aln pa; uint32_t order; } tmp_pair;
/* decreasing ln_prob (locs.rs:795); ties keep push order */
static int cmp_pair(const void* x, const void* y) {
    const tmp_pair* a = (const tmp_pair*)x; const tmp_pair* b = (const tmp_pair*)y;
    if (a->pa.ln_prob != b->pa.ln_prob) return a->pa.ln_prob > b->pa.ln_prob ? -1 : 1;
    return a->order < b->order ? -1 : (a->order > b->order ? 1 : 0);
}

typedef struct {
    lcty_pair_aln* v; size_t n, cap;
    tmp_pair* tmp; size_t tmp_cap;
    double* buffer; size_t buf_cap;
} pair_vec;

static void pv_push(pair_vec* pv, const lcty_pair_aln* pa) {
    if (pv->n == pv->cap) {
        pv->cap = pv->cap ? pv->cap * 2 : 256;
        pv->v = (lcty_pair_aln*)realloc(pv->v, sizeof(lcty_pair_aln) * pv->cap);
    }
    pv->v[pv->n++] = *pa;
}

/* identify_contig_pair_alns — src/model/locs.rs:746-799 */
static void identify_contig_pair_alns(const orc_locus* l, const o_aln* alns, size_t i, size_t j, size_t k,
                                      pair_vec* out, size_t max_alns, double unm_ins_penalty, double prob_diff) {
    size_t n2 = k - j;
    if (out->buf_cap < n2 + 1) { out->buf_cap = (n2 + 1) * 2; out->buffer = (double*)realloc(out->buffer, sizeof(double) * out->buf_cap); }
    size_t need = (j - i) * n2 + (j - i) + n2 + 1;
    if (out->tmp_cap < need) { out->tmp_cap = need * 2; out->tmp = (tmp_pair*)realloc(out->tmp, sizeof(tmp_pair) * out->tmp_cap); }
    double* buffer = out->buffer;
    for (size_t t = 0; t < n2; t++) buffer[t] = -INFINITY;
    tmp_pair* tmp = out->tmp; size_t nt = 0;
    for (size_t ix1 = i; ix1 < j; ix1++) {
        const o_aln* a1 = &alns[ix1];
        double max_prob1 = -INFINITY;
        for (size_t ix2 = j; ix2 < k; ix2++) {
            const o_aln* a2 = &alns[ix2];
            if (a1->reverse != a2->reverse) {
                /* paired_prob — aln.rs:236-238; furthest_distance — interv.rs:179-185 */
                uint32_t insert = MAX(a1->end, a2->end) - MIN(a1->start, a2->start);
                double prob = a1->ln_prob + a2->ln_prob + orc_locus_insert_lnprob(l, insert);
                if (isfinite(prob)) {
                    max_prob1 = fmax(max_prob1, prob);
                    buffer[ix2 - j] = fmax(buffer[ix2 - j], prob);
                    tmp_pair* tp = &tmp[nt]; memset(tp, 0, sizeof(*tp));
                    tp->pa.ln_prob = prob; tp->pa.contig = a1->contig;
                    tp->pa.ix1 = a1->rec_ix; tp->pa.mid1 = (a1->start + a1->end) / 2;
                    tp->pa.ix2 = a2->rec_ix; tp->pa.mid2 = (a2->start + a2->end) / 2;
                    tp->order = (uint32_t)nt; nt++;
                }
            }
        }
        double alone1 = a1->ln_prob + unm_ins_penalty;
        if (alone1 >= max_prob1) {
            tmp_pair* tp = &tmp[nt]; memset(tp, 0, sizeof(*tp));
            tp->pa.ln_prob = alone1; tp->pa.contig = a1->contig;
            tp->pa.ix1 = a1->rec_ix; tp->pa.mid1 = (a1->start + a1->end) / 2;
            tp->pa.ix2 = LCTY_NONE_U32; tp->pa.mid2 = LCTY_NONE_U32;
            tp->order = (uint32_t)nt; nt++;
        }
    }
    for (size_t ix2 = j; ix2 < k; ix2++) {
        const o_aln* a2 = &alns[ix2];
        double alone2 = a2->ln_prob + unm_ins_penalty;
        if (alone2 >= buffer[ix2 - j]) {
            tmp_pair* tp = &tmp[nt]; memset(tp, 0, sizeof(*tp));
            tp->pa.ln_prob = alone2; tp->pa.contig = a2->contig;
            tp->pa.ix1 = LCTY_NONE_U32; tp->pa.mid1 = LCTY_NONE_U32;
            tp->pa.ix2 = a2->rec_ix; tp->pa.mid2 = (a2->start + a2->end) / 2;
            tp->order = (uint32_t)nt; nt++;
        }
    }
    qsort(tmp, nt, sizeof(tmp_pair), cmp_pair);
    double thresh = tmp[0].pa.ln_prob - prob_diff;
    size_t lim = MIN(nt, max_alns), keep = 0;
    while (keep < lim && tmp[keep].pa.ln_prob >= thresh) keep++;    /* partition_point on a sorted slice */
    for (size_t t = 0; t < keep; t++) pv_push(out, &tmp[t].pa);
}

typedef struct {
    o_aln* kept; size_t cap;
} kept_vec;

/* ContigInfos::explicit_read_weight — model/windows.rs:683-693: the mean over the pair alignments of the larger of the two
 * read-end weights; 1.0 without explicit weights */
static double explicit_read_weight(const orc_locus* l, const lcty_pair_aln* pairs, size_t n) {
    if (!l->has_explicit) return 1.0;
    double s = 0.0;
    for (size_t t = 0; t < n; t++) {
        const double w1 = orc_read_end_weight(l, pairs[t].contig, pairs[t].mid1);
        const double w2 = orc_read_end_weight(l, pairs[t].contig, pairs[t].mid2);
        s += fmax(w1, w2);
    }
    return s / (double)n;
}

/* identify_paired_end_alignments — src/model/locs.rs:805-868 */
static void identify_paired_end(const orc_locus* l, prelim* p, size_t max_alns, double read_weight,
                                pair_vec* out, kept_vec* kv, double* weight_out, double* unmapped_out) {
    double insert_penalty = l->ins_mode_prob;
    double unm_ins_penalty = l->prm.unmapped_penalty + insert_penalty;
    qsort(p->alns, p->n_alns, sizeof(o_aln), cmp_pe);
    if (kv->cap < p->n_alns + 1) { kv->cap = (p->n_alns + 1) * 2; kv->kept = (o_aln*)realloc(kv->kept, sizeof(o_aln) * kv->cap); }
    o_aln* alignments = kv->kept; size_t k = 0;
    uint32_t curr_contig = 0;
    size_t i = 0, j = SIZE_MAX;
    size_t start_pairs = out->n;
    for (size_t t = p->n_alns; t-- > 0;) {     /* while let Some(aln) = tmp_alns.pop() */
        const o_aln* aln = &p->alns[t];
        if (curr_contig != aln->contig) {
            if (i < k)
                identify_contig_pair_alns(l, alignments, i, MIN(j, k), k, out, max_alns, unm_ins_penalty, l->prm.prob_diff);
            curr_contig = aln->contig;
            i = k; j = SIZE_MAX;
        }
        if (aln->read_end == 0) {
            if (k - i < max_alns) alignments[k++] = *aln;
        } else {
            j = MIN(j, k);
            if (k - j < max_alns) alignments[k++] = *aln;
        }
    }
    if (i < k)
        identify_contig_pair_alns(l, alignments, i, MIN(j, k), k, out, max_alns, unm_ins_penalty, l->prm.prob_diff);
    double weight = read_weight * explicit_read_weight(l, out->v + start_pairs, out->n - start_pairs);   /* locs.rs:860 */
    for (size_t t = start_pairs; t < out->n; t++) out->v[t].ln_prob *= weight;
    *weight_out = weight;
    *unmapped_out = weight * (2.0 * l->prm.unmapped_penalty + insert_penalty);
}

/* identify_single_end_alignments — src/model/locs.rs:873-911 */
static void identify_single_end(const orc_locus* l, prelim* p, size_t max_alns, double read_weight,
                                pair_vec* out, double* weight_out, double* unmapped_out) {
    qsort(p->alns, p->n_alns, sizeof(o_aln), cmp_se);
    int have_contig = 0; uint32_t curr_contig = 0;
    double thresh = NAN; size_t curr_saved = 0;
    size_t start_pairs = out->n;
    for (size_t t = p->n_alns; t-- > 0;) {
        const o_aln* aln = &p->alns[t];
        if (!have_contig || curr_contig != aln->contig) {
            have_contig = 1; curr_contig = aln->contig;
            thresh = aln->ln_prob - l->prm.prob_diff;
            curr_saved = 0;
        }
        if (aln->ln_prob >= thresh && curr_saved < max_alns) {
            lcty_pair_aln pa; memset(&pa, 0, sizeof(pa));
            pa.ln_prob = aln->ln_prob; pa.contig = aln->contig;
            pa.ix1 = aln->rec_ix; pa.mid1 = (aln->start + aln->end) / 2;
            pa.ix2 = LCTY_NONE_U32; pa.mid2 = LCTY_NONE_U32;
            pv_push(out, &pa);
            curr_saved++;
        }
    }
    double weight = read_weight * explicit_read_weight(l, out->v + start_pairs, out->n - start_pairs);   /* locs.rs:903 */
    for (size_t t = start_pairs; t < out->n; t++) out->v[t].ln_prob *= weight;
    *weight_out = weight;
    *unmapped_out = weight * l->prm.unmapped_penalty;
}

/* PosCollection::get — locs.rs:245-262 (the neighbour test as written: a hit when the stored start is 64 or more away) */
static int pos_get(const prelim* p, int read_end, uint32_t contig, uint32_t pos, uint32_t* index) {
    const uint64_t key = ((uint64_t)(read_end + 1) << 48) | ((uint64_t)contig << 32) | (uint64_t)(pos >> 7);
    for (int pass = 0; pass < 2; pass++) {
        const uint64_t kk = pass == 0 ? key : (uint64_t)((int64_t)key + ((pos & 64u) == 0 ? -1 : 1));
        size_t h = (size_t)mix64(kk) & (p->hcap - 1);
        while (p->hgen[h] == p->gen) {
            const pos_entry* e = &p->pos[p->hslot[h]];
            if (e->key == kk) {
                if (pass == 0) { *index = e->index; return 1; }
                const uint32_t d = e->pos > pos ? e->pos - pos : pos - e->pos;
                if ((d >> 6) != 0) { *index = e->index; return 1; }
                return 0;
            }
            h = (h + 1) & (p->hcap - 1);
        }
    }
    return 0;
}

static void revcomp(const uint8_t* in, uint32_t n, uint8_t* out) {
    for (uint32_t i = 0; i < n; i++) {
        const uint8_t c = in[n - 1 - i];
        out[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c;
    }
}

/* HapAlns::transfer_alignments — seq/transfer.rs:70-140. Returns the number of new alignments. */
static size_t transfer_alignments(load_ctx* c, const orc_hap_alns* hap, prelim* p, uint64_t pair, const uint8_t* const mate_seq[2],
                                  const uint8_t* const mate_rc[2], const int mate_rev[2]) {
    const orc_locus* l = c->l;
    const lcty_reads_host* in = c->in;
    const size_t n = p->n_alns;
    uint8_t* seen = (uint8_t*)calloc(n ? n : 1, 1);
    orc_cigar src, out; orc_cigar_init(&src); orc_cigar_init(&out);
    const uint32_t* cig_base = in->cigar + in->cigar_off[pair];
    const lcty_aln_rec* recs = in->recs + in->aln_off[pair];
    for (size_t i = 0; i < n; i++) {
        if (seen[i]) continue;
        seen[i] = 1;
        const o_aln sa = p->alns[i];                                     /* copied before the vector may change */
        if (sa.own_cigar) orc_cigar_from_raw(&src, sa.own_cigar, sa.own_n, 0);
        else orc_cigar_from_raw(&src, cig_base + recs[sa.rec_ix].cigar_rel, recs[sa.rec_ix].n_cigar, 1);
        const int e = sa.read_end;
        const uint32_t read_len = in->mate_len[2 * pair + e];
        const uint8_t* read_seq = (sa.reverse != 0) == (mate_rev[e] != 0) ? mate_seq[e] : mate_rc[e];     /* MateData::get_seq */
        const uint32_t passable = p->passable_dist[e];
        uint32_t fails_left = orc_hap_alns_transfer_fails(hap);
        const uint32_t nb = orc_hap_alns_n_best(hap, sa.contig);
        for (uint32_t t = 0; t < nb; t++) {
            const uint32_t target = orc_hap_alns_best(hap, sa.contig, t);
            const uint32_t approx = orc_hap_alns_approx_pos(hap, sa.contig, target, sa.start);
            uint32_t ix;
            if (pos_get(p, e, target, approx, &ix)) {                     /* a similar position is already there */
                if (ix < n) seen[ix] = 1;
                continue;
            }
            const uint32_t tlen = l->infos[target].len;
            const uint32_t new_start = orc_hap_alns_transfer(hap, sa.contig, target, sa.start, &src, read_seq, read_len, l->seqs[target], tlen, &out);
            const uint32_t ref_len = out.rlen, qlen = out.qlen;
            const uint32_t diff = ref_len > qlen ? ref_len - qlen : qlen - ref_len;
            if (diff > passable || ref_len < 50) {                        /* MIN_ALN_SIZE */
                if (fails_left == 0) break;
                fails_left--;
                continue;
            }
            /* Alignment::new + PrelimAlignments::push */
            uint32_t* raw = (uint32_t*)malloc(sizeof(uint32_t) * (out.n ? out.n : 1));
            orc_cigar_to_raw(&out, raw);
            if (p->n_owned == p->cap_owned) { p->cap_owned = p->cap_owned ? 2 * p->cap_owned : 16; p->owned = (uint32_t**)realloc(p->owned, sizeof(uint32_t*) * p->cap_owned); }
            p->owned[p->n_owned++] = raw;
            lcty_aln_rec fake; memset(&fake, 0, sizeof(fake));
            fake.pos = new_start; fake.contig = (uint16_t)target; fake.flags = (uint16_t)(LCTY_FLAG_SECONDARY | (sa.reverse ? LCTY_FLAG_REVERSE : 0));
            fake.n_cigar = out.n;
            o_aln na; int empty;
            if (!score_record(c, &fake, raw, 0, (uint8_t)e, &na, &empty) || empty) continue;
            /* ties between equally likely alignments keep the order of the merged table the product builds: the input records of
             * a read end first, then its transferred alignments in push order (the reference's sort_unstable leaves ties open) */
            na.rec_ix = 0x80000000u | (uint32_t)(p->n_owned - 1); na.own_cigar = raw; na.own_n = out.n;
            prelim_push(p, &na);
        }
    }
    orc_cigar_free(&src); orc_cigar_free(&out);
    free(seen);
    return p->n_alns - n;
}

static orc_alns* load_impl(const orc_locus* l, const lcty_reads_host* in, const orc_hap_alns* hap, int* err);
orc_alns* orc_load(const orc_locus* l, const lcty_reads_host* in, int* err) { return load_impl(l, in, NULL, err); }
orc_alns* orc_load_recover(const orc_locus* l, const lcty_reads_host* in, const orc_hap_alns* hap, int* err) { return load_impl(l, in, hap, err); }

/* Scratch of one thread of the load: sequence buffers, the output vector of its PairAlignments. */
typedef struct {
    uint8_t* seqbuf; uint8_t* mate_buf[4]; orc_u128* kbuf;
    pair_vec pv; kept_vec kv;
} load_scratch;

static void scratch_init(load_scratch* s, uint32_t max_len) {
    memset(s, 0, sizeof(*s));
    s->seqbuf = (uint8_t*)malloc(max_len);
    for (int t = 0; t < 4; t++) s->mate_buf[t] = (uint8_t*)malloc(max_len);
    s->kbuf = (orc_u128*)malloc(sizeof(orc_u128) * ((size_t)max_len + 1));
}
static void scratch_free(load_scratch* s) {
    for (int t = 0; t < 4; t++) free(s->mate_buf[t]);
    free(s->seqbuf); free(s->kbuf); free(s->pv.tmp); free(s->pv.buffer); free(s->kv.kept);
}
static void prelim_free(prelim* p) {
    for (size_t i = 0; i < p->n_owned; i++) free(p->owned[i]);
    free(p->owned); free(p->alns); free(p->pos); free(p->hslot); free(p->hgen);
}

/* The part of AllAlignments::load that runs inside the single-threaded BAM loop (locs.rs:1116-1150): read_next_alns for both
 * ends, in_bounds, calculate_read_weight. Returns 1 when the read goes on to recover_and_group_alignments with *weight. */
static int load_serial_part(load_ctx* c, orc_alns* A, uint64_t r, prelim* p, load_scratch* sc, int with_hap, double* weight_out) {
    const orc_locus* l = c->l; const lcty_reads_host* in = c->in;
    const int is_paired = l->bg.is_paired;
    const uint32_t boundary = l->prm.boundary_size - (uint32_t)l->prm.tweak;     /* locs.rs:1099 */
    uint64_t ri = in->aln_off[r], r_end = in->aln_off[r + 1];
    double weight = 1.0;
    prelim_reset(p, (size_t)(r_end - ri) + (with_hap ? (size_t)(r_end - ri) * l->n_alleles + l->n_alleles : 0));
    int well_mapped = read_next_alns(c, r, &ri, r_end, 0, &weight, p);    /* locs.rs:1119 */
    if (c->err) return 0;
    if (is_paired && well_mapped)
        well_mapped = read_next_alns(c, r, &ri, r_end, 1, &weight, p);    /* locs.rs:1125-1132 */
    if (c->err) return 0;
    if (!well_mapped) { A->status[r] = LCTY_READ_POORLY_MAPPED; return 0; }
    /* in_bounds — locs.rs:1008-1014 */
    int inb = 0;
    for (size_t t = 0; t < p->n_alns; t++) {
        uint32_t clen = l->infos[p->alns[t].contig].len;
        uint32_t mid = (p->alns[t].start + p->alns[t].end) / 2;
        if (boundary <= mid && mid < clen - boundary) { inb = 1; break; }
    }
    if (!inb) { A->status[r] = LCTY_READ_OUT_OF_BOUNDS; return 0; }
    /* calculate_read_weight — locs.rs:968-1002 */
    uint16_t paired_count = 0;
    for (int e = 0; e < 2; e++) {
        uint32_t len = in->mate_len[2 * r + e];
        if (len == 0) continue;
        if (e == 1 && !is_paired) continue;     /* mates[1] is None for single-end input */
        unpack_mate(in, 2 * r + e, sc->seqbuf);
        uint16_t cnt = count_unique_kmers(l, sc->seqbuf, len, sc->kbuf);
        A->uniq_kmers[2 * r + e] = cnt;
        paired_count = (uint16_t)(paired_count + cnt);
    }
    double w = l->weight_interc + (double)paired_count * l->weight_mult;
    w = w < 0.0 ? 0.0 : (w > 1.0 ? 1.0 : w);
    *weight_out = weight * w;
    return 1;
}

/* recover_and_group_alignments for one read (locs.rs:1255-1286): the part the reference runs on `threads` threads.
 * The read's PairAlignments are appended to sc->pv. */
static void load_group_part(load_ctx* c, orc_alns* A, uint64_t r, prelim* p, load_scratch* sc, const orc_hap_alns* hap, double weight,
                            uint64_t* n_good) {
    const orc_locus* l = c->l; const lcty_reads_host* in = c->in;
    const int is_paired = l->bg.is_paired;
    if (hap && weight >= l->prm.min_weight) {                        /* locs.rs:1257-1260 */
        const uint8_t* ms[2] = {NULL, NULL}; const uint8_t* mr[2] = {NULL, NULL};
        int mrev[2] = {0, 0};
        for (int e = 0; e < (is_paired ? 2 : 1); e++) {
            const uint32_t len = in->mate_len[2 * r + e];
            unpack_mate(in, 2 * r + e, sc->mate_buf[e]);
            revcomp(sc->mate_buf[e], len, sc->mate_buf[2 + e]);
            ms[e] = sc->mate_buf[e]; mr[e] = sc->mate_buf[2 + e];
        }
        /* strand of the mate's primary record (MateData::new) */
        {
            uint64_t q = in->aln_off[r];
            mrev[0] = (in->recs[q].flags & LCTY_FLAG_REVERSE) != 0;
            for (q = q + 1; q < in->aln_off[r + 1]; q++)
                if (!(in->recs[q].flags & (LCTY_FLAG_SECONDARY | LCTY_FLAG_SUPPL))) { mrev[1] = (in->recs[q].flags & LCTY_FLAG_REVERSE) != 0; break; }
        }
        transfer_alignments(c, hap, p, r, ms, mr, mrev);
        if (c->err) return;
    }
    if (!(p->best_edit[0] <= p->good_dist[0] && p->best_edit[1] <= p->good_dist[1])) {
        /* the read is dropped entirely; its MateData is unobservable -> reported as 0 */
        A->uniq_kmers[2 * r] = A->uniq_kmers[2 * r + 1] = 0;
        A->status[r] = LCTY_READ_POORLY_MAPPED; return;
    }
    for (size_t t = 0; t < p->n_alns; t++) p->alns[t].ln_prob -= p->best_lik[p->alns[t].read_end];   /* normalize_probs */
    size_t max_alns = weight >= l->prm.min_weight ? LCTY_MAX_USED_ALNS : LCTY_MAX_UNUSED_ALNS;
    double gw, unm;
    if (is_paired) identify_paired_end(l, p, max_alns, weight, &sc->pv, &sc->kv, &gw, &unm);
    else identify_single_end(l, p, max_alns, weight, &sc->pv, &gw, &unm);
    A->weight[r] = gw; A->unmapped_prob[r] = unm;
    if (gw >= l->prm.min_weight) { A->status[r] = LCTY_READ_GOOD; (*n_good)++; }
    else A->status[r] = LCTY_READ_FEW_KMERS;
}

static orc_alns* alns_alloc(const orc_locus* l, uint64_t R) {
    orc_alns* A = (orc_alns*)calloc(1, sizeof(orc_alns));
    A->n_pairs = R; A->n_alleles = l->n_alleles;
    A->status = (uint8_t*)calloc(R ? R : 1, 1);
    A->weight = (double*)calloc(R ? R : 1, sizeof(double));
    A->unmapped_prob = (double*)calloc(R ? R : 1, sizeof(double));
    A->uniq_kmers = (uint16_t*)calloc(R ? 2 * R : 1, sizeof(uint16_t));
    A->pa_off = (uint64_t*)calloc(R + 1, sizeof(uint64_t));
    return A;
}

static orc_alns* load_impl(const orc_locus* l, const lcty_reads_host* in, const orc_hap_alns* hap, int* err) {
    uint64_t R = in->n_pairs;
    orc_alns* A = alns_alloc(l, R);
    load_ctx c; c.l = l; c.in = in; c.err = 0;
    uint32_t boundary = l->prm.boundary_size - (uint32_t)l->prm.tweak;     /* locs.rs:1099 */
    for (uint32_t a = 0; a < l->n_alleles; a++)
        if (!(l->infos[a].len > 2 * boundary)) { c.err = LCTY_ERR_RUNTIME; }   /* assert! locs.rs:1100 */

    prelim p; memset(&p, 0, sizeof(p));
    uint32_t max_len = 1;
    for (uint64_t m = 0; m < 2 * R; m++) max_len = MAX(max_len, in->mate_len[m]);
    load_scratch sc; scratch_init(&sc, max_len);
    uint64_t n_good = 0;
    for (uint64_t r = 0; r < R && !c.err; r++) {
        A->pa_off[r] = sc.pv.n;
        double weight;
        if (!load_serial_part(&c, A, r, &p, &sc, hap != NULL, &weight)) continue;
        load_group_part(&c, A, r, &p, &sc, hap, weight, &n_good);
    }
    A->n_good = n_good;
    prelim_free(&p);
    pair_vec pv = sc.pv; sc.pv.v = NULL;
    scratch_free(&sc);
    if (c.err) {
        if (err) *err = c.err;
        free(pv.v); orc_alns_free(A);
        return NULL;
    }
    A->pa_off[R] = pv.n;
    A->pa = pv.v; A->n_pa = pv.n;
    if (err) *err = 0;
    return A;
}

/* ---- the same load with the reference's thread structure (the CPU baseline of bench.py) --------------------------------------
 * AllAlignments::load reads the BAM on one thread (locs.rs:1116-1150: read_next_alns, in_bounds, calculate_read_weight), deals the
 * reads that pass round-robin to `threads` vectors (1149) and runs recover_and_group_alignments on one thread per vector
 * (1157-1174). Here the input is processed in blocks of reads so that the PrelimAlignments of a block, not of the whole input,
 * are alive at a time: serial part of the block on the calling thread, then the block's reads grouped by the workers, the
 * k-th passing read on worker k % threads. The sum of the serial and of the parallel time is what the reference's two phases
 * take; results are identical to orc_load (per-read outputs do not depend on which thread produced them).
 * seconds[0] / seconds[1] (optional) receive the wall time of the serial / the grouping phases. */
#include <pthread.h>
#include <time.h>

typedef struct mt_shared mt_shared;
typedef struct {
    mt_shared* sh; uint32_t tid;
    load_scratch sc; load_ctx c;
    uint64_t n_good;
} mt_worker;
struct mt_shared {
    const orc_locus* l; const lcty_reads_host* in; const orc_hap_alns* hap; orc_alns* A;
    uint32_t threads, block;
    prelim* prelims;             /* [block] */
    uint64_t* read_of;           /* [block] read index of the k-th passing read of the block */
    double* weight_of;           /* [block] */
    uint32_t n_pass; uint64_t first_k;   /* passing reads in this block; number of passing reads before it */
    uint32_t* pa_thread; uint64_t* pa_start; uint32_t* pa_count;   /* [R] where a read's PairAlignments are */
    pthread_barrier_t go, done;
    int stop;
};

static void* mt_worker_main(void* arg) {
    mt_worker* w = (mt_worker*)arg; mt_shared* sh = w->sh;
    for (;;) {
        pthread_barrier_wait(&sh->go);
        if (sh->stop) break;
        for (uint32_t k = 0; k < sh->n_pass; k++) {
            if ((sh->first_k + k) % sh->threads != w->tid) continue;         /* locs.rs:1149 */
            const uint64_t r = sh->read_of[k];
            const size_t before = w->sc.pv.n;
            if (!w->c.err) load_group_part(&w->c, sh->A, r, &sh->prelims[k], &w->sc, sh->hap, sh->weight_of[k], &w->n_good);
            sh->pa_thread[r] = w->tid; sh->pa_start[r] = before; sh->pa_count[r] = (uint32_t)(w->sc.pv.n - before);
        }
        pthread_barrier_wait(&sh->done);
    }
    return NULL;
}

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

orc_alns* orc_load_mt(const orc_locus* l, const lcty_reads_host* in, const orc_hap_alns* hap, uint32_t threads, double* seconds, int* err) {
    if (threads < 1) threads = 1;
    const uint64_t R = in->n_pairs;
    orc_alns* A = alns_alloc(l, R);
    load_ctx c; c.l = l; c.in = in; c.err = 0;
    uint32_t boundary = l->prm.boundary_size - (uint32_t)l->prm.tweak;
    for (uint32_t a = 0; a < l->n_alleles; a++)
        if (!(l->infos[a].len > 2 * boundary)) { c.err = LCTY_ERR_RUNTIME; }
    uint32_t max_len = 1;
    for (uint64_t m = 0; m < 2 * R; m++) max_len = MAX(max_len, in->mate_len[m]);
    mt_shared sh; memset(&sh, 0, sizeof(sh));
    sh.l = l; sh.in = in; sh.hap = hap; sh.A = A; sh.threads = threads;
    sh.block = MAX(64u, 32u * threads);
    sh.prelims = (prelim*)calloc(sh.block, sizeof(prelim));
    sh.read_of = (uint64_t*)malloc(sizeof(uint64_t) * sh.block);
    sh.weight_of = (double*)malloc(sizeof(double) * sh.block);
    sh.pa_thread = (uint32_t*)calloc(R ? R : 1, sizeof(uint32_t));
    sh.pa_start = (uint64_t*)calloc(R ? R : 1, sizeof(uint64_t));
    sh.pa_count = (uint32_t*)calloc(R ? R : 1, sizeof(uint32_t));
    pthread_barrier_init(&sh.go, NULL, threads + 1);
    pthread_barrier_init(&sh.done, NULL, threads + 1);
    mt_worker* ws = (mt_worker*)calloc(threads, sizeof(mt_worker));
    pthread_t* tids = (pthread_t*)calloc(threads, sizeof(pthread_t));
    for (uint32_t t = 0; t < threads; t++) {
        ws[t].sh = &sh; ws[t].tid = t; ws[t].c = c; scratch_init(&ws[t].sc, max_len);
        pthread_create(&tids[t], NULL, mt_worker_main, &ws[t]);
    }
    load_scratch sc; scratch_init(&sc, max_len);
    double t_serial = 0.0, t_group = 0.0;
    uint64_t r = 0;
    while (r < R && !c.err) {
        double t0 = now_s();
        sh.n_pass = 0;
        while (r < R && sh.n_pass < sh.block && !c.err) {
            double weight;
            if (load_serial_part(&c, A, r, &sh.prelims[sh.n_pass], &sc, hap != NULL, &weight)) {
                sh.read_of[sh.n_pass] = r; sh.weight_of[sh.n_pass] = weight; sh.n_pass++;
            }
            r++;
        }
        double t1 = now_s();
        pthread_barrier_wait(&sh.go);
        pthread_barrier_wait(&sh.done);
        sh.first_k += sh.n_pass;
        t_serial += t1 - t0; t_group += now_s() - t1;
    }
    sh.stop = 1;
    pthread_barrier_wait(&sh.go);
    for (uint32_t t = 0; t < threads; t++) { pthread_join(tids[t], NULL); if (ws[t].c.err && !c.err) c.err = ws[t].c.err; }
    /* AllAlignments::extend (locs.rs:1168-1172, 1187-1190): the threads' results are concatenated; here in input order */
    double t2 = now_s();
    uint64_t total = 0;
    for (uint64_t q = 0; q < R; q++) total += sh.pa_count[q];
    A->pa = (lcty_pair_aln*)malloc(sizeof(lcty_pair_aln) * (total ? total : 1));
    uint64_t at = 0;
    for (uint64_t q = 0; q < R; q++) {
        A->pa_off[q] = at;
        if (sh.pa_count[q]) memcpy(A->pa + at, ws[sh.pa_thread[q]].sc.pv.v + sh.pa_start[q], sizeof(lcty_pair_aln) * sh.pa_count[q]);
        at += sh.pa_count[q];
    }
    A->pa_off[R] = at; A->n_pa = at;
    t_group += now_s() - t2;
    for (uint32_t t = 0; t < threads; t++) { A->n_good += ws[t].n_good; free(ws[t].sc.pv.v); scratch_free(&ws[t].sc); }
    for (uint32_t k = 0; k < sh.block; k++) prelim_free(&sh.prelims[k]);
    scratch_free(&sc); free(sc.pv.v);
    free(sh.prelims); free(sh.read_of); free(sh.weight_of); free(sh.pa_thread); free(sh.pa_start); free(sh.pa_count);
    pthread_barrier_destroy(&sh.go); pthread_barrier_destroy(&sh.done);
    free(ws); free(tids);
    if (seconds) { seconds[0] = t_serial; seconds[1] = t_group; }
    if (c.err) { if (err) *err = c.err; orc_alns_free(A); return NULL; }
    if (err) *err = 0;
    return A;
}

/* test hook: one Cigar::transfer_read_alignment behind HapAlns */
uint32_t orc_transfer_one(const orc_hap_alns* h, uint32_t source, uint32_t target, uint32_t source_start, const uint32_t* read_cigar,
                          uint32_t n_read_cigar, const uint8_t* read_seq, uint32_t read_len, const uint8_t* target_seq,
                          uint32_t target_len, uint32_t* out_cigar, uint32_t out_cap, uint32_t* n_out) {
    orc_cigar src, out; orc_cigar_init(&src); orc_cigar_init(&out);
    orc_cigar_from_raw(&src, read_cigar, n_read_cigar, 1);
    const uint32_t st = orc_hap_alns_transfer(h, source, target, source_start, &src, read_seq, read_len, target_seq, target_len, &out);
    *n_out = out.n;
    if (out.n <= out_cap) orc_cigar_to_raw(&out, out_cigar);
    orc_cigar_free(&src); orc_cigar_free(&out);
    return st;
}

/* Test hook: an orc_alns assembled from arrays (e.g. the products of the GPU scoring kernel), so that the solver
 * stages of oracle and GPU can be compared on bit-identical inputs. */
orc_alns* orc_alns_from_arrays(uint64_t n_pairs, uint32_t n_alleles, const uint8_t* status, const double* weight,
                               const double* unmapped_prob, const uint64_t* pa_off, const lcty_pair_aln* pa) {
    orc_alns* A = (orc_alns*)calloc(1, sizeof(orc_alns));
    A->n_pairs = n_pairs; A->n_alleles = n_alleles;
    A->status = (uint8_t*)malloc(n_pairs ? n_pairs : 1); memcpy(A->status, status, n_pairs);
    A->weight = (double*)malloc(sizeof(double) * (n_pairs ? n_pairs : 1)); memcpy(A->weight, weight, sizeof(double) * n_pairs);
    A->unmapped_prob = (double*)malloc(sizeof(double) * (n_pairs ? n_pairs : 1)); memcpy(A->unmapped_prob, unmapped_prob, sizeof(double) * n_pairs);
    A->uniq_kmers = (uint16_t*)calloc(n_pairs ? 2 * n_pairs : 1, sizeof(uint16_t));
    A->pa_off = (uint64_t*)malloc(sizeof(uint64_t) * (n_pairs + 1)); memcpy(A->pa_off, pa_off, sizeof(uint64_t) * (n_pairs + 1));
    A->n_pa = pa_off[n_pairs];
    A->pa = (lcty_pair_aln*)malloc(sizeof(lcty_pair_aln) * (A->n_pa ? A->n_pa : 1)); memcpy(A->pa, pa, sizeof(lcty_pair_aln) * A->n_pa);
    for (uint64_t r = 0; r < n_pairs; r++) A->n_good += status[r] == LCTY_READ_GOOD;
    return A;
}

void orc_alns_free(orc_alns* a) {
    if (!a) return;
    free(a->status); free(a->weight); free(a->unmapped_prob); free(a->uniq_kmers); free(a->pa_off); free(a->pa);
    free(a);
}
uint64_t orc_alns_n_pairs(const orc_alns* a) { return a->n_pairs; }
uint64_t orc_alns_n_good(const orc_alns* a) { return a->n_good; }
void orc_alns_status(const orc_alns* a, uint8_t* status, double* weight, double* unmapped_prob, uint16_t* uniq_kmers) {
    if (status) memcpy(status, a->status, a->n_pairs);
    if (weight) memcpy(weight, a->weight, sizeof(double) * a->n_pairs);
    if (unmapped_prob) memcpy(unmapped_prob, a->unmapped_prob, sizeof(double) * a->n_pairs);
    if (uniq_kmers) memcpy(uniq_kmers, a->uniq_kmers, sizeof(uint16_t) * 2 * a->n_pairs);
}
uint64_t orc_alns_pair_alns(const orc_alns* a, uint64_t* off, lcty_pair_aln* out, uint64_t cap) {
    if (off) memcpy(off, a->pa_off, sizeof(uint64_t) * (a->n_pairs + 1));
    if (out && a->n_pa && cap) memcpy(out, a->pa, sizeof(lcty_pair_aln) * MIN((uint64_t)a->n_pa, cap));
    return a->n_pa;
}

/* best_aln_matrix — locs.rs:1203-1212 via best_for_each_contig 621-629 */
void orc_best_aln_matrix(const orc_alns* a, double* out) {
    uint64_t j = 0, ng = a->n_good;
    for (uint64_t r = 0; r < a->n_pairs; r++) {
        if (a->status[r] != LCTY_READ_GOOD) continue;
        uint64_t t = a->pa_off[r], te = a->pa_off[r + 1];
        for (uint32_t c = 0; c < a->n_alleles; c++) {
            double v = a->unmapped_prob[r];
            if (t < te && a->pa[t].contig == c) {
                v = a->pa[t].ln_prob;
                while (t < te && a->pa[t].contig == c) t++;
            }
            out[(uint64_t)c * ng + j] = v;
        }
        j++;
    }
}

/* ======================================================================== */
/* genotypes / prefilter                                                     */
/* ======================================================================== */

/* count_combinations — src/ext/vec.rs:285-296 */
uint64_t orc_count_genotypes(uint32_t n_alleles, uint32_t ploidy) {
    uint64_t n = (uint64_t)n_alleles + ploidy - 1, r = ploidy;
    if (r > n) return 0;
    uint64_t m = MIN(r, n - r), acc = 1;
    for (uint64_t v = 1; v <= m; v++) acc = acc * (n - v + 1) / v;
    return acc;
}

/* gen_combinations_with_repl — src/ext/vec.rs:298-339 */
static void rec_comb(uint32_t n, uint16_t* buffer, uint32_t start, uint32_t depth, uint32_t size,
                     uint16_t* out, uint64_t* w) {
    if (depth + 1 == size) {
        for (uint32_t el = start; el < n; el++) {
            buffer[depth] = (uint16_t)el;
            memcpy(out + (*w) * size, buffer, sizeof(uint16_t) * size);
            (*w)++;
        }
    } else {
        for (uint32_t el = start; el < n; el++) {
            buffer[depth] = (uint16_t)el;
            rec_comb(n, buffer, el, depth + 1, size, out, w);
        }
    }
}
uint64_t orc_generate_genotypes(uint32_t n_alleles, uint32_t ploidy, uint16_t* out) {
    uint64_t w = 0;
    if (n_alleles == 0 || ploidy == 0) return 0;
    uint16_t buffer[64];
    if (ploidy > 64) return 0;
    rec_comb(n_alleles, buffer, 0, 0, ploidy, out, &w);
    return w;
}

/* run_filter — src/solvers/solve.rs:101-119 */
void orc_run_filter(const double* matrix, uint32_t n_alleles, uint64_t n_good,
                    const uint16_t* genotypes, uint64_t n_gt, uint32_t ploidy,
                    const double* priors, double* scores) {
    (void)n_alleles;
    double* best = (double*)malloc(sizeof(double) * (n_good ? n_good : 1));
    for (uint64_t g = 0; g < n_gt; g++) {
        const uint16_t* ids = genotypes + g * ploidy;
        memcpy(best, matrix + (uint64_t)ids[0] * n_good, sizeof(double) * n_good);
        for (uint32_t t = 1; t < ploidy; t++) {
            const double* row = matrix + (uint64_t)ids[t] * n_good;
            for (uint64_t r = 0; r < n_good; r++) best[r] = fmax(best[r], row[r]);
        }
        double s = -0.0;      /* f64 Sum::sum folds from -0.0 */
        for (uint64_t r = 0; r < n_good; r++) s += best[r];
        scores[g] = (priors ? priors[g] : 0.0) + s;
    }
    free(best);
}

typedef struct { double score; uint64_t ix; } sc_ix;
static int cmp_score(const void* x, const void* y) {
    const sc_ix* a = (const sc_ix*)x; const sc_ix* b = (const sc_ix*)y;
    if (a->score != b->score) return a->score > b->score ? -1 : 1;   /* total_cmp on non-NaN scores */
    return a->ix < b->ix ? -1 : (a->ix > b->ix ? 1 : 0);
}

/* truncate_ixs — src/solvers/solve.rs:52-84 */
uint64_t orc_truncate(const double* scores, uint64_t* ixs, uint64_t n, double filt_diff,
                      uint64_t min_size, uint64_t threads) {
    if (n == 0) return 0;
    sc_ix* v = (sc_ix*)malloc(sizeof(sc_ix) * n);
    for (uint64_t i = 0; i < n; i++) { v[i].ix = ixs[i]; v[i].score = scores[ixs[i]]; }
    qsort(v, n, sizeof(sc_ix), cmp_score);
    for (uint64_t i = 0; i < n; i++) ixs[i] = v[i].ix;
    double best = v[0].score, worst = v[n - 1].score;
    double thresh = best - filt_diff;
    uint64_t m = n;
    if (!(min_size >= n || worst >= thresh)) {
        m = 0; while (m < n && v[m].score >= thresh) m++;
        if (m < min_size) {
            thresh = v[min_size - 1].score;
            m = 0; while (m < n && v[m].score >= thresh) m++;
        }
        m = MAX(m, threads); m = MIN(m, n);
    }
    free(v);
    return m;
}
