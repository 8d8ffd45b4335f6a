/*
 * lcty_oracle_solve.c — CPU restatement of the solver stages of `locityper genotype`
 * (SURVEY.md §8a rows a24-a33): GenotypeWindows / GenotypeAlignments, apply_tweak, ReadAssignment,
 * Greedy, SimAnneal, the stage loop and the final genotype comparison.
 * TEST INFRASTRUCTURE ONLY — see lcty_oracle.h. PARITY UNPINNED.
 *
 * Randomness: the reference drives everything from one Xoshiro256++ through rand ^0.10 adaptors that
 * are not in the tree and whose results already depend on `--threads` (solve.rs:1017, 1051). Here
 * every (genotype, attempt) chain is driven by one caller-supplied 64-bit seed:
 *   - apply_tweak uses counter-based draws orc_counter_u64(seed, i) (order-free, so that the GPU can
 *     evaluate them in parallel),
 *   - the solver uses xoshiro256++ seeded by seed_from_u64(seed) with the adaptors documented in
 *     lcty_oracle.h. The GPU kernels consume exactly the same definitions.
 */
#include "lcty_oracle_internal.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MIN(a, b) ((a) < (b) ? (a) : (b))
#define MAX(a, b) ((a) > (b) ? (a) : (b))

/* ---------------------------------------------------------------- RNG */
static inline uint64_t splitmix64_next(uint64_t* x) {
    uint64_t z = (*x += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

void orc_rng_seed(orc_rng* r, uint64_t seed) {         /* rand_core seed_from_u64 for xoshiro: SplitMix64 fill */
    for (int i = 0; i < 4; i++) r->s[i] = splitmix64_next(&seed);
}
uint64_t orc_rng_next(orc_rng* r) {                     /* xoshiro256++ */
    uint64_t* s = r->s;
    const uint64_t result = rotl64(s[0] + s[3], 23) + s[0];
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl64(s[3], 45);
    return result;
}
static void rng_jump_with(orc_rng* r, const uint64_t poly[4]) {
    uint64_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int i = 0; i < 4; i++)
        for (int b = 0; b < 64; b++) {
            if (poly[i] & (1ULL << b)) { s0 ^= r->s[0]; s1 ^= r->s[1]; s2 ^= r->s[2]; s3 ^= r->s[3]; }
            orc_rng_next(r);
        }
    r->s[0] = s0; r->s[1] = s1; r->s[2] = s2; r->s[3] = s3;
}
void orc_rng_jump(orc_rng* r) {
    static const uint64_t J[4] = {0x180ec6d33cfd0abaULL, 0xd5a61266f0c9392cULL, 0xa9582618e03fc9aaULL, 0x39abdc4529b1661cULL};
    rng_jump_with(r, J);
}
void orc_rng_long_jump(orc_rng* r) {
    static const uint64_t J[4] = {0x76e15d3efefdcbbfULL, 0xc5004e441c522fb3ULL, 0x77710069854ee241ULL, 0x39109bb02acbe635ULL};
    rng_jump_with(r, J);
}
uint64_t orc_rng_below(orc_rng* r, uint64_t n) {
    return (uint64_t)(((unsigned __int128)orc_rng_next(r) * (unsigned __int128)n) >> 64);
}
double orc_rng_f64(orc_rng* r) { return (double)(orc_rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }
uint64_t orc_counter_u64(uint64_t key, uint64_t i) {
    uint64_t z = key + (i + 1) * 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
#define WINDOW_KEY_XOR 0xD1B54A32D192ED03ULL

/* ---------------------------------------------------------------- window weights / distributions */
/* WeightCalculator::{new,get} — model/windows.rs:163-177 */
double orc_weight_calc(double breakpoint, double power, double x) {
    const double const_fct = pow(breakpoint / (1.0 - breakpoint), power);
    return 1.0 / (1.0 + const_fct * pow((1.0 - x) / x, power));
}

/* ContigInfo::neighb_info — windows.rs:439-445 */
double orc_window_weight(const orc_locus* l, uint32_t allele, uint32_t wstart, uint32_t* gc) {
    const orc_contig_info* ci = &l->infos[allele];
    const uint32_t i = wstart > l->left_padding ? wstart - l->left_padding : 0;
    if (gc) *gc = ci->gc[i];
    if (l->win_weight_inj) return l->win_weight_inj[l->ci_off_inj[allele] + i];
    const double uniq_frac = (double)ci->uniq_cnt[i] * l->uniq_mult;          /* windows.rs:402 */
    const double complexity = (double)ci->compl_cnt[i] * l->compl_mult;      /* compl.rs:133,138 */
    double w = 1.0;
    if (l->prm.kmers_weight_bp > 0.0) w = orc_weight_calc(l->prm.kmers_weight_bp, l->prm.kmers_weight_pow, uniq_frac);
    if (l->prm.compl_weight_bp > 0.0) w = w * orc_weight_calc(l->prm.compl_weight_bp, l->prm.compl_weight_pow, complexity);
    /* info.explicit_weight: 1.0 (windows.rs:336) or the average over the window itself (409-413) */
    const double ew = l->has_explicit ? orc_explicit_average(l, allele, i + l->left_padding, i + l->left_padding + l->bg.window) : 1.0;
    return w * ew;
}

/* every per-position window weight of the locus (alleles concatenated, len - neighb + 1 values each): the oracle's own values,
 * whatever has been injected; and BayesCalc::ln_pmf for gc 0..100 x depth lo..hi-1 (out[gc * (hi - lo) + d - lo]) */
void orc_locus_window_weights(const orc_locus* l, double* out) {
    double* inj = l->win_weight_inj;
    ((orc_locus*)l)->win_weight_inj = NULL;
    size_t at = 0;
    for (uint32_t a = 0; a < l->n_alleles; a++) {
        const uint32_t npos = l->infos[a].len + 1 - l->bg.neighb;
        for (uint32_t i = 0; i < npos; i++) out[at++] = orc_window_weight(l, a, i + l->left_padding, NULL);
    }
    ((orc_locus*)l)->win_weight_inj = inj;
}
void orc_depth_table(const lcty_bg* bg, const lcty_params* prm, uint32_t lo, uint32_t hi, double* out) {
    for (uint32_t gc = 0; gc < LCTY_GC_BINS; gc++)
        for (uint32_t d = lo; d < hi; d++) out[(size_t)gc * (hi - lo) + (d - lo)] = orc_depth_ln_pmf(bg, prm, gc, d);
}

/* WindowDistr::ln_prob — distr_cache.rs:34-39 with LinearCache (lincache.rs:41-48) */
double orc_depth_ln_prob(const orc_locus* l, uint32_t gc, double weight, uint32_t depth) {
    if (weight == 0.0) return 0.0;                 /* WindowDistr::TRIVIAL */
    const double v = depth < LCTY_DEPTH_CACHE ? l->depth_lut[gc * LCTY_DEPTH_CACHE + depth]
                   : depth < l->depth_ext_width ? l->depth_ext_inj[(size_t)gc * l->depth_ext_width + depth]
                                              : orc_depth_ln_pmf(&l->bg, &l->prm, gc, depth);
    return weight * v;
}

/* ---------------------------------------------------------------- GenotypeAlignments */
typedef struct {
    double ln_prob;
    uint32_t mid1, mid2;
    uint32_t win[2];
    uint8_t contig_ix;        /* 0xFF: both mates unmapped (parent == None) */
} gt_aln;

struct orc_gt_alns {
    const orc_locus* l;
    uint32_t ploidy;
    uint16_t ids[16];
    uint32_t wshifts[17];
    uint64_t n_reads;
    gt_aln* alns; uint64_t n_alns;
    uint64_t* read_ixs;       /* [n_reads + 1] */
    uint64_t* non_trivial; uint64_t n_nontrivial;
    uint32_t total_windows;
    uint8_t* w_gc; double* w_weight;          /* depth_distrs after apply_tweak */
    double depth_contrib, aln_contrib;
    /* capacities of the arrays above: a worker of orc_solve_stage refills ONE object for the genotypes of its run (orc_gt_alns_fill)
     * instead of allocating ~100 B per read pair and genotype anew — 128 workers doing that at once spend their time in page faults */
    uint64_t cap_alns, cap_reads; uint32_t cap_windows;
};

typedef struct { gt_aln a; uint32_t order; } tmp_loc;
static int cmp_loc(const void* x, const void* y) {
    const tmp_loc* a = (const tmp_loc*)x; const tmp_loc* b = (const tmp_loc*)y;
    if (a->a.ln_prob != b->a.ln_prob) return a->a.ln_prob > b->a.ln_prob ? -1 : 1;     /* windows.rs:793, ties: push order */
    return a->order < b->order ? -1 : (a->order > b->order ? 1 : 0);
}

/* GenotypeAlignments::new (assgn.rs:41-84) into `g`, whose arrays are kept and grown as needed */
static void orc_gt_alns_fill(orc_gt_alns* g, const orc_locus* l, const orc_alns* A, const uint16_t* ids, uint32_t ploidy) {
    g->l = l; g->ploidy = ploidy;
    g->n_alns = 0; g->n_nontrivial = 0;
    /* GenotypeWindows::new — windows.rs:721-739 */
    uint32_t shift = 2;                                     /* REG_WINDOW_SHIFT */
    g->wshifts[0] = shift;
    for (uint32_t i = 0; i < ploidy; i++) {
        g->ids[i] = ids[i];
        shift += l->infos[ids[i]].n_windows;
        g->wshifts[i + 1] = shift;
    }
    g->total_windows = shift;
    if (shift > g->cap_windows) {
        free(g->w_gc); free(g->w_weight);
        g->w_gc = (uint8_t*)malloc(shift); g->w_weight = (double*)malloc(sizeof(double) * shift);
        g->cap_windows = shift;
    }
    memset(g->w_gc, 0, shift); memset(g->w_weight, 0, sizeof(double) * shift);
    g->aln_contrib = 1.0 - l->prm.lik_skew;                 /* assgn.rs:80-81 */
    g->depth_contrib = 1.0 + l->prm.lik_skew;
    g->n_reads = A->n_good;
    if (A->n_good + 1 > g->cap_reads) {
        free(g->read_ixs); free(g->non_trivial);
        g->read_ixs = (uint64_t*)malloc(sizeof(uint64_t) * (A->n_good + 1));
        g->non_trivial = (uint64_t*)malloc(sizeof(uint64_t) * (A->n_good ? A->n_good : 1));
        g->cap_reads = A->n_good + 1;
    }
    g->read_ixs[0] = 0;
    uint64_t cap = g->cap_alns;
    if (cap < A->n_good * (ploidy + 1) + 16) {
        cap = A->n_good * (ploidy + 1) + 16;
        free(g->alns);
        g->alns = (gt_aln*)malloc(sizeof(gt_aln) * cap);
    }
    tmp_loc* tmp = NULL; size_t tmp_cap = 0;
    const double prob_diff = l->prm.prob_diff;
    uint64_t rp = 0;
    for (uint64_t r = 0; r < A->n_pairs; r++) {
        if (A->status[r] != LCTY_READ_GOOD) continue;
        /* extend_read_gt_alns — windows.rs:762-797 */
        const double unmapped_prob = A->unmapped_prob[r];
        double thresh = unmapped_prob - prob_diff;
        const uint64_t lo = A->pa_off[r], hi = A->pa_off[r + 1];
        size_t nt = 0;
        if (tmp_cap < (size_t)(hi - lo) * ploidy + 2) { tmp_cap = ((size_t)(hi - lo) * ploidy + 2) * 2; tmp = (tmp_loc*)realloc(tmp, sizeof(tmp_loc) * tmp_cap); }
        for (uint32_t i = 0; i < ploidy; i++) {
            /* contig_alns: the run of pair alignments on contig ids[i] (sorted contig asc, ln_prob desc) */
            uint64_t s = lo;
            while (s < hi && A->pa[s].contig < ids[i]) s++;
            uint64_t e = s;
            while (e < hi && A->pa[e].contig == ids[i]) e++;
            if (s < e) {
                thresh = fmax(thresh, A->pa[s].ln_prob - prob_diff);
                for (uint64_t t = s; t < e; t++) {
                    if (A->pa[t].ln_prob >= thresh) {
                        tmp_loc* x = &tmp[nt];
                        x->a.ln_prob = A->pa[t].ln_prob; x->a.mid1 = A->pa[t].mid1; x->a.mid2 = A->pa[t].mid2;
                        x->a.contig_ix = (uint8_t)i; x->a.win[0] = x->a.win[1] = 0; x->order = (uint32_t)nt; nt++;
                    } else break;
                }
            }
        }
        if (unmapped_prob >= thresh) {
            tmp_loc* x = &tmp[nt];
            x->a.ln_prob = unmapped_prob; x->a.mid1 = x->a.mid2 = LCTY_NONE_U32; x->a.contig_ix = 0xFF;
            x->a.win[0] = x->a.win[1] = 0; x->order = (uint32_t)nt; nt++;
        }
        qsort(tmp, nt, sizeof(tmp_loc), cmp_loc);
        size_t keep = 0;
        while (keep < nt && tmp[keep].a.ln_prob >= thresh) keep++;
        if (g->n_alns + keep > cap) { cap = (g->n_alns + keep) * 2; g->alns = (gt_aln*)realloc(g->alns, sizeof(gt_aln) * cap); }
        for (size_t t = 0; t < keep; t++) g->alns[g->n_alns++] = tmp[t].a;
        g->read_ixs[rp + 1] = g->n_alns;
        if (keep > 1) g->non_trivial[g->n_nontrivial++] = rp;       /* assgn.rs:61-63 */
        rp++;
    }
    free(tmp);
    g->cap_alns = cap;
}

orc_gt_alns* orc_gt_alns_new(const orc_locus* l, const orc_alns* A, const uint16_t* ids, uint32_t ploidy) {
    if (ploidy == 0 || ploidy > 16) return NULL;
    orc_gt_alns* g = (orc_gt_alns*)calloc(1, sizeof(*g));
    orc_gt_alns_fill(g, l, A, ids, ploidy);
    return g;
}

void orc_gt_alns_free(orc_gt_alns* g) {
    if (!g) return;
    free(g->alns); free(g->read_ixs); free(g->non_trivial); free(g->w_gc); free(g->w_weight); free(g);
}
uint64_t orc_gt_alns_n_reads(const orc_gt_alns* g) { return g->n_reads; }
uint64_t orc_gt_alns_n_alns(const orc_gt_alns* g) { return g->n_alns; }
uint32_t orc_gt_alns_n_windows(const orc_gt_alns* g) { return g->total_windows; }
uint64_t orc_gt_alns_n_nontrivial(const orc_gt_alns* g) { return g->n_nontrivial; }

void orc_gt_alns_get(const orc_gt_alns* g, uint64_t* read_ixs, double* ln_prob, uint8_t* contig_ix,
                     uint32_t* mid1, uint32_t* mid2, uint32_t* windows, uint64_t* non_trivial) {
    if (read_ixs) memcpy(read_ixs, g->read_ixs, sizeof(uint64_t) * (g->n_reads + 1));
    for (uint64_t i = 0; i < g->n_alns; i++) {
        if (ln_prob) ln_prob[i] = g->alns[i].ln_prob;
        if (contig_ix) contig_ix[i] = g->alns[i].contig_ix;
        if (mid1) mid1[i] = g->alns[i].mid1;
        if (mid2) mid2[i] = g->alns[i].mid2;
        if (windows) { windows[2 * i] = g->alns[i].win[0]; windows[2 * i + 1] = g->alns[i].win[1]; }
    }
    if (non_trivial) memcpy(non_trivial, g->non_trivial, sizeof(uint64_t) * g->n_nontrivial);
}

/* get_shifted_window_ix + WindowGetter::middle_window — windows.rs:62-68, 465-470 */
static inline uint32_t shifted_window_ix(const orc_contig_info* ci, uint32_t window, uint32_t shift, uint32_t middle) {
    if (middle == LCTY_NONE_U32) return 0;                            /* UNMAPPED_WINDOW */
    if (ci->reg_start <= middle && middle < ci->reg_end) return (middle - ci->reg_start) / window + shift;
    return 1;                                                         /* BOUNDARY_WINDOW */
}

/* apply_tweak — assgn.rs:127-151; define_windows_{determ,random} windows.rs:112-136; generate_windows 478-486 */
void orc_gt_alns_apply_tweak(orc_gt_alns* g, uint64_t key) {
    const orc_locus* l = g->l;
    const uint32_t tweak = (uint32_t)l->prm.tweak, window = l->bg.window;
    for (uint64_t rp = 0; rp < g->n_reads; rp++)
    for (uint64_t i = g->read_ixs[rp]; i < g->read_ixs[rp + 1]; i++) {
        gt_aln* a = &g->alns[i];
        if (a->contig_ix == 0xFF) { a->win[0] = a->win[1] = 0; continue; }   /* parent == None: windows stay UNMAPPED */
        const orc_contig_info* ci = &l->infos[g->ids[a->contig_ix]];
        const uint32_t shift = g->wshifts[a->contig_ix];
        uint32_t t1 = 0, t2 = 0;
        if (tweak != 0) {
            /* one draw per location, keyed by (read, location index inside the read) */
            const uint64_t r = orc_counter_u64(key, (rp << 16) | (i - g->read_ixs[rp]));
            t1 = (uint32_t)(r >> 32) % (2 * tweak + 1);
            t2 = (uint32_t)r % (2 * tweak + 1);
        }
        a->win[0] = shifted_window_ix(ci, window, shift, a->mid1 == LCTY_NONE_U32 ? LCTY_NONE_U32 : a->mid1 + t1);
        a->win[1] = shifted_window_ix(ci, window, shift, a->mid2 == LCTY_NONE_U32 ? LCTY_NONE_U32 : a->mid2 + t2);
    }
    g->w_weight[0] = g->w_weight[1] = 0.0;                            /* two TRIVIAL distributions, assgn.rs:75-77 */
    g->w_gc[0] = g->w_gc[1] = 0;
    uint32_t w = 2;
    for (uint32_t i = 0; i < g->ploidy; i++) {
        const orc_contig_info* ci = &l->infos[g->ids[i]];
        for (uint32_t j = 0; j < ci->n_windows; j++, w++) {
            const uint32_t start = ci->reg_start + j * window, end = start + window;
            const uint32_t left = MIN(tweak, start), right = MIN(tweak, ci->len - end);
            /* rng.random_range(-left..=right) */
            const uint64_t r = orc_counter_u64(key ^ WINDOW_KEY_XOR, w);
            const int64_t off = -(int64_t)left + (int64_t)(((unsigned __int128)r * (unsigned __int128)(left + right + 1)) >> 64);
            const uint32_t wstart = (uint32_t)((int64_t)start + off);
            uint32_t gc;
            const double weight = orc_window_weight(l, g->ids[i], wstart, &gc);
            if (weight < l->prm.min_weight || weight < 1e-7) { g->w_weight[w] = 0.0; g->w_gc[w] = 0; }   /* assgn.rs:144-148, distr_cache.rs:84 */
            else { g->w_weight[w] = weight; g->w_gc[w] = (uint8_t)gc; }
        }
    }
}

void orc_gt_alns_window_distr(const orc_gt_alns* g, uint8_t* gc, double* weight) {
    if (gc) memcpy(gc, g->w_gc, g->total_windows);
    if (weight) memcpy(weight, g->w_weight, sizeof(double) * g->total_windows);
}

/* max_aln_lik — assgn.rs:160-163 */
double orc_gt_alns_max_aln_lik(const orc_gt_alns* g) {
    double s = -0.0;
    for (uint64_t r = 0; r < g->n_reads; r++) s += g->alns[g->read_ixs[r]].ln_prob;
    return s;
}

/* ---------------------------------------------------------------- ReadAssignment (assgn.rs:171-426) */
typedef struct {
    const orc_gt_alns* g;
    uint16_t* assgn;
    uint32_t* depth;
    double aln_lik, depth_lik;
} rassgn;

static inline double window_ln_prob(const orc_gt_alns* g, uint32_t w, uint32_t depth) {
    return orc_depth_ln_prob(g->l, g->w_gc[w], g->w_weight[w], depth);
}

/* recalc_likelihood — assgn.rs:346-354 */
static void recalc_likelihood(rassgn* ra) {
    const orc_gt_alns* g = ra->g;
    double s = -0.0;
    for (uint32_t w = 0; w < g->total_windows; w++) s += window_ln_prob(g, w, ra->depth[w]);
    ra->depth_lik = s;
    s = -0.0;
    for (uint64_t r = 0; r < g->n_reads; r++) s += g->alns[g->read_ixs[r] + ra->assgn[r]].ln_prob;
    ra->aln_lik = s;
}

/* try_new — assgn.rs:199-226; init: 0 = best location, 1 = random_range(0..alns.len()) */
#define INIT_KEY_XOR 0x8CB92BA72F3D8DD7ULL
static void rassgn_init(rassgn* ra, const orc_gt_alns* g, int random_init, uint64_t key) {
    ra->g = g;
    ra->assgn = (uint16_t*)calloc(g->n_reads ? g->n_reads : 1, sizeof(uint16_t));
    ra->depth = (uint32_t*)calloc(g->total_windows, sizeof(uint32_t));
    for (uint64_t r = 0; r < g->n_reads; r++) {
        const uint64_t i = g->read_ixs[r], m = g->read_ixs[r + 1] - i;
        uint32_t a = 0;
        /* rng.random_range(0..alns.len()) per non-trivial read -> counter draw keyed by the read (order-free) */
        if (m > 1 && random_init)
            a = (uint32_t)(((unsigned __int128)orc_counter_u64(key ^ INIT_KEY_XOR, r) * (unsigned __int128)m) >> 64);
        ra->assgn[r] = (uint16_t)a;
        ra->depth[g->alns[i + a].win[0]]++;
        ra->depth[g->alns[i + a].win[1]]++;
    }
    recalc_likelihood(ra);
}
static void rassgn_free(rassgn* ra) { free(ra->assgn); free(ra->depth); }

static inline double rassgn_likelihood(const rassgn* ra) {        /* assgn.rs:235-237 */
    return ra->g->depth_contrib * ra->depth_lik + ra->g->aln_contrib * ra->aln_lik;
}

/* atomic_depth_lik_diff — assgn.rs:244-254 */
static inline double atomic_diff(const rassgn* ra, uint32_t w, int32_t change) {
    if (change == 0) return 0.0;
    const uint32_t old_depth = ra->depth[w];
    const uint32_t new_depth = (uint32_t)((int64_t)old_depth + change);
    return window_ln_prob(ra->g, w, new_depth) - window_ln_prob(ra->g, w, old_depth);
}

/* depth_lik_diff — assgn.rs:259-284 */
static double depth_lik_diff(const rassgn* ra, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4) {
    int32_t c1 = -1, c2, c3, c4;
    if (w2 == w1) { c1 -= 1; c2 = 0; } else c2 = -1;
    if (w3 == w1) { c1 += 1; c3 = 0; } else if (w3 == w2) { c2 += 1; c3 = 0; } else c3 = 1;
    if (w4 == w1) { c1 += 1; c4 = 0; } else if (w4 == w2) { c2 += 1; c4 = 0; } else if (w4 == w3) { c3 += 1; c4 = 0; } else c4 = 1;
    return atomic_diff(ra, w1, c1) + atomic_diff(ra, w2, c2) + atomic_diff(ra, w3, c3) + atomic_diff(ra, w4, c4);
}

typedef struct { uint64_t read_pair; uint16_t new_assgn; uint64_t old_ix, new_ix; } target;

/* best_read_improvement — assgn.rs:287-317 */
static double best_read_improvement(const rassgn* ra, uint64_t rp, target* out) {
    const orc_gt_alns* g = ra->g;
    const uint64_t start = g->read_ixs[rp], end = g->read_ixs[rp + 1];
    const uint16_t old_assgn = ra->assgn[rp];
    const uint64_t old_ix = start + old_assgn;
    const gt_aln* old_aln = &g->alns[old_ix];
    uint64_t best_i = 0;
    double best_improv = -INFINITY;
    const double rel_contrib = g->depth_contrib / g->aln_contrib;
    for (uint64_t i = 0; i < end - start; i++) {
        if (i == old_assgn) continue;
        const gt_aln* a = &g->alns[start + i];
        const double improv = a->ln_prob + rel_contrib * depth_lik_diff(ra, old_aln->win[0], old_aln->win[1], a->win[0], a->win[1]);
        if (improv > best_improv) { best_improv = improv; best_i = i; }
    }
    out->read_pair = rp; out->new_assgn = (uint16_t)best_i; out->old_ix = old_ix; out->new_ix = start + best_i;
    return g->aln_contrib * (best_improv - old_aln->ln_prob);
}

/* calculate_improvement — assgn.rs:321-328 */
static double calculate_improvement(const rassgn* ra, const target* t) {
    const orc_gt_alns* g = ra->g;
    const gt_aln* o = &g->alns[t->old_ix]; const gt_aln* n = &g->alns[t->new_ix];
    return g->depth_contrib * depth_lik_diff(ra, o->win[0], o->win[1], n->win[0], n->win[1])
           + g->aln_contrib * (n->ln_prob - o->ln_prob);
}

/* reassign — assgn.rs:331-343 */
static void reassign(rassgn* ra, const target* t) {
    const orc_gt_alns* g = ra->g;
    const gt_aln* o = &g->alns[t->old_ix]; const gt_aln* n = &g->alns[t->new_ix];
    ra->depth_lik += depth_lik_diff(ra, o->win[0], o->win[1], n->win[0], n->win[1]);
    ra->aln_lik += n->ln_prob - o->ln_prob;
    ra->depth[n->win[0]] += 1; ra->depth[n->win[1]] += 1;
    ra->depth[o->win[0]] -= 1; ra->depth[o->win[1]] -= 1;
    ra->assgn[t->read_pair] = t->new_assgn;
}

/* ReassignmentTarget::random — assgn.rs:451-471 */
static void target_random(const rassgn* ra, orc_rng* rng, target* t) {
    const orc_gt_alns* g = ra->g;
    const uint64_t rp = g->non_trivial[orc_rng_below(rng, g->n_nontrivial)];
    const uint64_t start = g->read_ixs[rp], total = g->read_ixs[rp + 1] - start;
    const uint16_t old_assgn = ra->assgn[rp];
    uint16_t new_assgn;
    if (total == 2) new_assgn = (uint16_t)(1 - old_assgn);
    else {
        const uint16_t i = (uint16_t)(1 + orc_rng_below(rng, total - 1));   /* random_range(1..total) */
        new_assgn = i <= old_assgn ? (uint16_t)(i - 1) : i;
    }
    t->read_pair = rp; t->new_assgn = new_assgn; t->old_ix = start + old_assgn; t->new_ix = start + new_assgn;
}

/* max_abs_random — stoch.rs:19-22 */
static double max_abs_random(const rassgn* ra, orc_rng* rng, int count) {
    double acc = 0.0;
    for (int i = 0; i < count; i++) {
        target t; target_random(ra, rng, &t);
        acc = fmax(acc, fabs(calculate_improvement(ra, &t)));
    }
    return acc;
}
static inline double minimum_allowed_diff(double m) { return fmax(1e-10 * m, 1e-14); }   /* stoch.rs:27-29 */

/* Greedy::solve_nontrivial — stoch.rs:81-120 */
static void solve_greedy(rassgn* ra, const orc_gt_alns* g, const lcty_solver* s, orc_rng* rng, uint64_t key) {
    const uint64_t nnt = g->n_nontrivial;
    const uint64_t sample_size = MIN((uint64_t)s->sample_size, nnt);
    rassgn_init(ra, g, !s->best_start, key);
    const double min_diff = minimum_allowed_diff(max_abs_random(ra, rng, 100));
    uint64_t curr_plato = 0;
    const uint64_t max_iter = MAX((uint64_t)100000, (uint64_t)s->plato_size * 100);
    uint64_t* picked = (uint64_t*)malloc(sizeof(uint64_t) * (sample_size ? sample_size : 1));
    for (uint64_t it = 0; it < max_iter; it++) {
        int have = 0; target best_t; memset(&best_t, 0, sizeof(best_t));
        double best_improv = min_diff;
        /* non_trivial_reads.sample(rng, sample_size) (our adaptor): one draw of the chain's generator per iteration; the picks are
         * counter draws under that key, taken in order, repeats skipped — distinct indices, and the picks of an iteration do not
         * depend on each other (the GPU draws them on separate lanes) */
        const uint64_t sample_key = orc_rng_next(rng);
        uint64_t ctr = 0;
        for (uint64_t j = 0; j < sample_size; j++) {
            uint64_t idx;
            int dup;
            do {
                idx = (uint64_t)(((unsigned __int128)orc_counter_u64(sample_key, ctr++) * (unsigned __int128)nnt) >> 64);
                dup = 0;
                for (uint64_t q = 0; q < j; q++) dup |= picked[q] == idx;
            } while (dup);
            picked[j] = idx;
            target t;
            const double improv = best_read_improvement(ra, g->non_trivial[idx], &t);
            if (improv > best_improv) { best_t = t; best_improv = improv; have = 1; }
        }
        if (have) { curr_plato = 0; reassign(ra, &best_t); }
        else { curr_plato++; if (curr_plato > s->plato_size) break; }
    }
    free(picked);
}

/* SimAnneal::solve_nontrivial — stoch.rs:195-245 */
static void solve_anneal(rassgn* ra, const orc_gt_alns* g, const lcty_solver* s, orc_rng* rng, uint64_t key) {
    rassgn_init(ra, g, 1, key);
    const double max_abs = max_abs_random(ra, rng, 100);
    const double min_diff = minimum_allowed_diff(max_abs);
    const double start_temp = fmax(-max_abs / log(s->init_prob), 1e-5);
    const double temp_step = start_temp / (double)s->anneal_steps;
    uint64_t curr_plato = 0;
    for (uint64_t i = s->anneal_steps; i >= 1; i--) {
        target t; target_random(ra, rng, &t);
        const double diff = calculate_improvement(ra, &t) - min_diff;
        if (diff >= 0.0 || orc_rng_f64(rng) <= exp(diff / (temp_step * (double)i))) { reassign(ra, &t); curr_plato = 0; }
        else { curr_plato++; if (curr_plato >= s->plato_size) break; }
    }
    const uint64_t max_iter = MAX((uint64_t)100000, (uint64_t)s->plato_size * 100);
    for (uint64_t it = 0; it < max_iter; it++) {
        if (curr_plato >= s->plato_size) break;
        target t; target_random(ra, rng, &t);
        const double diff = calculate_improvement(ra, &t);
        if (diff > min_diff) { reassign(ra, &t); curr_plato = 0; } else curr_plato++;
    }
}

/* The optimum of the integer programme the reference hands to HiGHS / Gurobi (src/solvers/highs.rs:38-100, gurobi.rs:15-83): one
 * binary per (non-trivial read, location) with objective aln_contrib * ln_prob, one-hot depth variables per window with objective
 * depth_contrib * ln_prob(depth), coupling rows. Its optimum is the assignment of largest ReadAssignment::likelihood
 * (assgn.rs:235-237); here by EXHAUSTIVE ENUMERATION of the non-trivial reads' locations, every assignment valued from scratch with
 * recalc_likelihood (assgn.rs:346-354). Test infrastructure for small models: more than max_states assignments -> the likelihood
 * is NaN. Ties: the first assignment in odometer order (read 0 fastest) of the largest value. */
static void solve_enumerate(rassgn* ra, const orc_gt_alns* g, uint64_t max_states) {
    rassgn_init(ra, g, 0, 0);
    double states = 1.0;
    for (uint64_t q = 0; q < g->n_nontrivial; q++) {
        const uint64_t r = g->non_trivial[q];
        states *= (double)(g->read_ixs[r + 1] - g->read_ixs[r]);
    }
    if (states > (double)max_states) { ra->aln_lik = NAN; ra->depth_lik = NAN; return; }
    uint16_t* best = (uint16_t*)malloc(sizeof(uint16_t) * (g->n_reads ? g->n_reads : 1));
    memcpy(best, ra->assgn, sizeof(uint16_t) * g->n_reads);
    double best_lik = rassgn_likelihood(ra);
    for (;;) {
        /* odometer over the non-trivial reads */
        uint64_t q = 0;
        for (; q < g->n_nontrivial; q++) {
            const uint64_t r = g->non_trivial[q], i = g->read_ixs[r], m = g->read_ixs[r + 1] - i;
            const gt_aln* old = &g->alns[i + ra->assgn[r]];
            ra->depth[old->win[0]]--; ra->depth[old->win[1]]--;
            ra->assgn[r] = (uint16_t)((uint64_t)ra->assgn[r] + 1 < m ? ra->assgn[r] + 1 : 0);
            const gt_aln* nw = &g->alns[i + ra->assgn[r]];
            ra->depth[nw->win[0]]++; ra->depth[nw->win[1]]++;
            if (ra->assgn[r] != 0) break;
        }
        if (q == g->n_nontrivial) break;                 /* wrapped around: every assignment seen */
        recalc_likelihood(ra);
        const double lik = rassgn_likelihood(ra);
        if (lik > best_lik) { best_lik = lik; memcpy(best, ra->assgn, sizeof(uint16_t) * g->n_reads); }
    }
    memset(ra->depth, 0, sizeof(uint32_t) * g->total_windows);
    memcpy(ra->assgn, best, sizeof(uint16_t) * g->n_reads);
    for (uint64_t r = 0; r < g->n_reads; r++) {
        const gt_aln* a = &g->alns[g->read_ixs[r] + ra->assgn[r]];
        ra->depth[a->win[0]]++; ra->depth[a->win[1]]++;
    }
    recalc_likelihood(ra);
    free(best);
}

void orc_solver_default(lcty_solver* s, int32_t kind) {
    memset(s, 0, sizeof(*s));
    s->kind = kind;
    s->best_start = 1; s->sample_size = 10;                         /* Greedy::default stoch.rs:45-52 */
    s->plato_size = kind == LCTY_SOLVER_GREEDY ? 100 : 10000;       /* SimAnneal::default stoch.rs:161-168 */
    s->anneal_steps = 20000; s->init_prob = 0.5;
}

/* Solver::solve — solvers/mod.rs:57-72 */
double orc_solve(const orc_gt_alns* g, const lcty_solver* s, uint64_t seed, uint16_t* assgn_out, double* lik_parts) {
    rassgn ra;
    orc_rng rng; orc_rng_seed(&rng, seed);
    if (g->n_nontrivial == 0) rassgn_init(&ra, g, 0, seed);         /* trivial: the only possible assignment */
    else if (s->kind == LCTY_SOLVER_GREEDY) solve_greedy(&ra, g, s, &rng, seed);
    else if (s->kind == LCTY_SOLVER_EXACT) solve_enumerate(&ra, g, s->node_limit ? s->node_limit : 4000000u);
    else solve_anneal(&ra, g, s, &rng, seed);
    const double lik = rassgn_likelihood(&ra);
    if (assgn_out) memcpy(assgn_out, ra.assgn, sizeof(uint16_t) * g->n_reads);
    if (lik_parts) { lik_parts[0] = ra.aln_lik; lik_parts[1] = ra.depth_lik; }
    rassgn_free(&ra);
    return lik;
}

double orc_assignment_likelihood(const orc_gt_alns* g, const uint16_t* assgn, double* lik_parts) {
    rassgn ra; ra.g = g;
    ra.assgn = (uint16_t*)malloc(sizeof(uint16_t) * (g->n_reads ? g->n_reads : 1));
    memcpy(ra.assgn, assgn, sizeof(uint16_t) * g->n_reads);
    ra.depth = (uint32_t*)calloc(g->total_windows, sizeof(uint32_t));
    for (uint64_t r = 0; r < g->n_reads; r++) {
        const gt_aln* a = &g->alns[g->read_ixs[r] + assgn[r]];
        ra.depth[a->win[0]]++; ra.depth[a->win[1]]++;
    }
    recalc_likelihood(&ra);
    const double lik = rassgn_likelihood(&ra);
    if (lik_parts) { lik_parts[0] = ra.aln_lik; lik_parts[1] = ra.depth_lik; }
    rassgn_free(&ra);
    return lik;
}

/* solve_single_thread inner loops — solve.rs:816-843 */
void orc_solve_stage(const orc_locus* l, const orc_alns* a, const uint16_t* genotypes, uint64_t n_gt, uint32_t ploidy,
                     const double* priors, const lcty_solver* s, uint32_t attempts, const uint64_t* chain_seeds,
                     double* lik_mean, double* lik_var, double* liks_out) {
    double* liks = (double*)malloc(sizeof(double) * (attempts ? attempts : 1));
    if (ploidy == 0 || ploidy > 16) { for (uint64_t gi = 0; gi < n_gt; gi++) lik_mean[gi] = lik_var[gi] = NAN; free(liks); return; }
    orc_gt_alns* g = (orc_gt_alns*)calloc(1, sizeof(*g));       /* this worker's GenotypeAlignments, refilled per genotype */
    for (uint64_t gi = 0; gi < n_gt; gi++) {
        orc_gt_alns_fill(g, l, a, genotypes + gi * ploidy, ploidy);
        const double prior = priors ? priors[gi] : 0.0;
        for (uint32_t at = 0; at < attempts; at++) {
            const uint64_t seed = chain_seeds[gi * attempts + at];
            orc_gt_alns_apply_tweak(g, seed);
            liks[at] = prior + orc_solve(g, s, seed, NULL, NULL);
            if (liks_out) liks_out[gi * attempts + at] = liks[at];
        }
        /* mean_variance_or_nan — ext/vec.rs:74-116 */
        double sum = -0.0;
        for (uint32_t at = 0; at < attempts; at++) sum += liks[at];
        const double mean = attempts ? sum / (double)attempts : NAN;
        double var = NAN;
        if (attempts > 1) {
            double acc = 0.0;
            for (uint32_t at = 0; at < attempts; at++) { const double d = liks[at] - mean; acc += d * d; }
            var = acc / (double)(attempts - 1);
        }
        lik_mean[gi] = mean; lik_var[gi] = var;
    }
    orc_gt_alns_free(g);
    free(liks);
}

/* The same stage with the reference's worker threads (MainWorker::run, solve.rs:1047-1062): the stage's genotype list is cut into
 * `threads` contiguous runs (ceil-div of what is left over the workers that are left), one run per worker; a worker handles its
 * genotypes one after the other (Worker::run, 1108-1146). The reference shuffles the list first (1051) only to balance the runs;
 * chains here are driven by their own seeds, so the result equals orc_solve_stage for any number of threads. */
#include <pthread.h>
typedef struct {
    const orc_locus* l; const orc_alns* a; const uint16_t* genotypes; uint64_t n_gt; uint32_t ploidy; const double* priors;
    const lcty_solver* s; uint32_t attempts; const uint64_t* chain_seeds; double* lik_mean; double* lik_var; double* liks_out;
} stage_task;
static void* stage_worker(void* arg) {
    stage_task* t = (stage_task*)arg;
    orc_solve_stage(t->l, t->a, t->genotypes, t->n_gt, t->ploidy, t->priors, t->s, t->attempts, t->chain_seeds, t->lik_mean, t->lik_var,
                    t->liks_out);
    return NULL;
}
void orc_solve_stage_mt(const orc_locus* l, const orc_alns* a, const uint16_t* genotypes, uint64_t n_gt, uint32_t ploidy,
                        const double* priors, const lcty_solver* s, uint32_t attempts, const uint64_t* chain_seeds,
                        double* lik_mean, double* lik_var, double* liks_out, uint32_t threads) {
    if (threads < 1) threads = 1;
    stage_task* tasks = (stage_task*)calloc(threads, sizeof(stage_task));
    pthread_t* tids = (pthread_t*)calloc(threads, sizeof(pthread_t));
    uint64_t start = 0; uint32_t used = 0;
    for (uint32_t i = 0; i < threads && start < n_gt; i++) {
        const uint64_t rem_workers = threads - i;
        const uint64_t jobs = (n_gt - start + rem_workers - 1) / rem_workers;         /* fast_ceil_div, solve.rs:1057 */
        stage_task* t = &tasks[used];
        t->l = l; t->a = a; t->genotypes = genotypes + start * ploidy; t->n_gt = jobs; t->ploidy = ploidy;
        t->priors = priors ? priors + start : NULL; t->s = s; t->attempts = attempts; t->chain_seeds = chain_seeds + start * attempts;
        t->lik_mean = lik_mean + start; t->lik_var = lik_var + start; t->liks_out = liks_out ? liks_out + start * attempts : NULL;
        pthread_create(&tids[used], NULL, stage_worker, t);
        used++; start += jobs;
    }
    for (uint32_t i = 0; i < used; i++) pthread_join(tids[i], NULL);
    free(tasks); free(tids);
}

/* GenotypeAlignments::create_counts + ReadAssignment::update_counts over the attempts of one genotype —
 * assgn.rs:94-96, 374-378 as driven by solve.rs:821-836. read_ixs_out[n_reads + 1], counts_out[n_alns] (or NULL to
 * size them). Returns n_alns. */
uint64_t orc_assignment_counts(const orc_locus* l, const orc_alns* a, const uint16_t* ids, uint32_t ploidy, const lcty_solver* s,
                               uint32_t attempts, const uint64_t* chain_seeds, uint64_t* read_ixs_out, uint16_t* counts_out) {
    orc_gt_alns* g = orc_gt_alns_new(l, a, ids, ploidy);
    const uint64_t n_alns = g->n_alns;
    if (read_ixs_out) for (uint64_t r = 0; r <= g->n_reads; r++) read_ixs_out[r] = g->read_ixs[r];
    if (counts_out) {
        memset(counts_out, 0, sizeof(uint16_t) * n_alns);
        uint16_t* assgn = (uint16_t*)malloc(sizeof(uint16_t) * (g->n_reads ? g->n_reads : 1));
        for (uint32_t at = 0; at < attempts; at++) {
            orc_gt_alns_apply_tweak(g, chain_seeds[at]);
            orc_solve(g, s, chain_seeds[at], assgn, NULL);
            for (uint64_t r = 0; r < g->n_reads; r++) counts_out[g->read_ixs[r] + assgn[r]] += 1;
        }
        free(assgn);
    }
    orc_gt_alns_free(g);
    return n_alns;
}

/* ---------------------------------------------------------------- K15: comparing genotypes */
/* compare_two_likelihoods — solve.rs:319-336 (DIFF_VAR = false selects the Welch branch: the const generic is
 * named EQ_VAR in math/mod.rs:180 and receives `false`) */
double orc_compare_two_likelihoods(double mean1, double var1, uint32_t att1, double mean2, double var2, uint32_t att2) {
    const double simple_norm = mean1 - orc_ln_add(mean1, mean2);
    const int n1 = isnormal(var1), n2 = isnormal(var2);
    if (n1 && n2) {
        const double t_pval = att1 == att2 ? orc_t_test(mean1, var1, mean2, var2, (double)att1)
                                           : orc_t_test_diffsizes(mean1, var1, mean2, var2, (double)att1, (double)att2);
        return fmax(simple_norm, log(t_pval));
    }
    return simple_norm;
}

typedef struct { double mean; uint64_t ix; } mean_ix;
static int cmp_mean(const void* x, const void* y) {
    const mean_ix* a = (const mean_ix*)x; const mean_ix* b = (const mean_ix*)y;
    if (a->mean != b->mean) return a->mean > b->mean ? -1 : 1;      /* sort_indices solve.rs:418-423; ties by index */
    return a->ix < b->ix ? -1 : (a->ix > b->ix ? 1 : 0);
}
static void sort_by_mean(const double* lik_mean, uint64_t* ixs, uint64_t n) {
    mean_ix* v = (mean_ix*)malloc(sizeof(mean_ix) * (n ? n : 1));
    for (uint64_t i = 0; i < n; i++) { v[i].ix = ixs[i]; v[i].mean = lik_mean[ixs[i]]; }
    qsort(v, n, sizeof(mean_ix), cmp_mean);
    for (uint64_t i = 0; i < n; i++) ixs[i] = v[i].ix;
    free(v);
}

/* discard_improbable_genotypes — solve.rs:425-480 */
uint64_t orc_discard_improbable(const double* lik_mean, const double* lik_var, const uint32_t* attempts, uint64_t* ixs,
                                uint64_t n, double prob_thresh, uint64_t out_size, uint64_t threads) {
    out_size = MAX(out_size, threads);
    if (prob_thresh == -INFINITY || out_size >= n) return n;
    sort_by_mean(lik_mean, ixs, n);
    const uint64_t best = ixs[0];
    uint64_t m = out_size;
    if (out_size <= 500) {                                          /* SOPHISTICATED_COUNT */
        uint32_t dropped = 0;
        for (uint64_t t = out_size; t < n; t++) {
            const uint64_t ix = ixs[t];
            const double ln_pval = orc_compare_two_likelihoods(lik_mean[ix], lik_var[ix], attempts[ix],
                                                               lik_mean[best], lik_var[best], attempts[best]);
            if (ln_pval >= prob_thresh) ixs[m++] = ix;
            else { dropped++; if (dropped >= 5) break; }            /* STOP_COUNT */
        }
    }
    return m;
}

/* produce_result — solve.rs:482-535 */
uint64_t orc_produce_result(const double* lik_mean, const double* lik_var, const uint32_t* attempts, const uint64_t* ixs_in,
                            uint64_t n_in, double prob_thresh, uint64_t out_bams, uint64_t* out_ixs, double* out_ln_probs,
                            double* quality) {
    const double THRESH = -11.512925464970229;
    const uint64_t MAX_GENOTYPES = 50;
    const uint64_t min_output = MAX((uint64_t)4, out_bams);
    const double thresh_prob = fmin(THRESH, prob_thresh);
    uint64_t* ixs = (uint64_t*)malloc(sizeof(uint64_t) * (n_in ? n_in : 1));
    memcpy(ixs, ixs_in, sizeof(uint64_t) * n_in);
    sort_by_mean(lik_mean, ixs, n_in);
    uint64_t n = MIN(n_in, MAX_GENOTYPES);
    double* ln_probs = (double*)calloc(n ? n : 1, sizeof(double));
    uint64_t i = 0;
    while (i < n) {
        const uint64_t u = ixs[i];
        for (uint64_t j = i + 1; j < n; j++) {
            const uint64_t v = ixs[j];
            const double prob_j = orc_compare_two_likelihoods(lik_mean[v], lik_var[v], attempts[v], lik_mean[u], lik_var[u], attempts[u]);
            if (i == 0 && j >= min_output && prob_j < thresh_prob) { n = j; break; }
            ln_probs[i] += log1p(-exp(prob_j));
            ln_probs[j] += prob_j;
        }
        i++;
    }
    const double norm = orc_ln_sum(ln_probs, n);
    for (uint64_t t = 0; t < n; t++) { ln_probs[t] -= norm; out_ixs[t] = ixs[t]; out_ln_probs[t] = ln_probs[t]; }
    /* Phred::from_ln_prob(Ln::sum(&ln_probs[1..])).min(1e9) — math/mod.rs:109-112 */
    const double q = -10.0 * (orc_ln_sum(ln_probs + 1, n > 0 ? n - 1 : 0) * 0.4342944819032518277);
    if (quality) *quality = fmin(q, 1e9);
    free(ixs); free(ln_probs);
    return n;
}

/* count_unexplained_reads — solve.rs:718-729 */
uint32_t orc_count_unexplained(const orc_alns* a, const uint16_t* ids, uint32_t ploidy) {
    uint32_t unexplained = 0;
    for (uint64_t r = 0; r < a->n_pairs; r++) {
        if (a->status[r] != LCTY_READ_GOOD) continue;
        double best = -INFINITY;
        for (uint32_t i = 0; i < ploidy; i++) {
            double v = a->unmapped_prob[r];                         /* best_at_contig — locs.rs:605-611 */
            for (uint64_t t = a->pa_off[r]; t < a->pa_off[r + 1]; t++)
                if (a->pa[t].contig == ids[i]) { v = a->pa[t].ln_prob; break; }
            best = fmax(best, v);
        }
        unexplained += best < a->unmapped_prob[r] + 1e-8;
    }
    return unexplained;
}

/* genotype_distance — solve.rs:339-357; gen_permutations — ext/vec.rs:342-372 (n >= 3: the loop only calls `action` after a
 * swap, so the initial order is never visited) */
static uint32_t perm_dist(const uint16_t* a, const uint16_t* b, uint32_t ploidy, const uint32_t* dist, uint32_t n_alleles) {
    uint32_t d = 0;
    for (uint32_t i = 0; i < ploidy; i++) {
        if (a[i] != b[i]) {
            const uint32_t v = dist[(size_t)a[i] * n_alleles + b[i]];
            if (v == 0xFFFFFFFFu) return 0xFFFFFFFFu;
            d += v;
        }
    }
    return d;
}
static uint32_t orc_genotype_distance(const uint16_t* gt1, const uint16_t* gt2, uint32_t n, const uint32_t* dist, uint32_t n_alleles) {
    uint32_t min_dist = 0xFFFFFFFFu, d;
    uint16_t buffer[16];
    memcpy(buffer, gt1, sizeof(uint16_t) * n);
    if (n == 1) {
        d = perm_dist(buffer, gt2, n, dist, n_alleles); if (d < min_dist) min_dist = d;
    } else if (n == 2) {
        d = perm_dist(buffer, gt2, n, dist, n_alleles); if (d < min_dist) min_dist = d;
        const uint16_t sw[2] = {gt1[1], gt1[0]};
        d = perm_dist(sw, gt2, n, dist, n_alleles); if (d < min_dist) min_dist = d;
    } else {
        uint32_t c[16] = {0};
        uint32_t i = 1;
        while (i < n) {
            if (c[i] < i) {
                const uint32_t j = c[i] * (i % 2);                       /* 0 if i is even, c[i] if i is odd */
                const uint16_t t = buffer[i]; buffer[i] = buffer[j]; buffer[j] = t;
                d = perm_dist(buffer, gt2, n, dist, n_alleles); if (d < min_dist) min_dist = d;
                c[i] += 1; i = 1;
            } else { c[i] = 0; i += 1; }
        }
    }
    return min_dist;
}

/* Genotyping::find_weighted_dist (solve.rs:621-636), check_first_prob (638-646), check_num_of_reads (650-675):
 * warnings bit 0 = NoProbableGenotype, bit 1 = FewReads; weighted_dist NaN = None */
void orc_call_checks(const uint16_t* genotypes, uint64_t n, uint32_t ploidy, const double* ln_probs, uint32_t n_reads,
                     const uint32_t* dist, uint32_t n_alleles, uint32_t* distances_out, double* weighted_dist, uint32_t* warnings) {
    uint32_t w = 0;
    if (isnan(ln_probs[0]) || ln_probs[0] < -2.0 * log(10.0)) w |= 1u;
    if (n_reads < ploidy) w |= 2u;
    else if (ploidy > 1 && n_reads < ploidy * 10) {
        const double k = (double)ploidy, nr = (double)n_reads;
        const double exp_zeros = exp(log(k - 1.0) * nr - log(k) * (nr - 1.0));
        if (exp_zeros > 0.1) w |= 2u;
    }
    *warnings = w;
    if (!dist) { *weighted_dist = NAN; return; }
    double sum_prob = 0.0, sum_dist = 0.0;
    int have = 1;
    for (uint64_t i = 0; i < n; i++) {
        const double prob = exp(ln_probs[i]);
        sum_prob += prob;
        const uint32_t d = i > 0 ? orc_genotype_distance(genotypes, genotypes + i * ploidy, ploidy, dist, n_alleles) : 0u;
        if (d == 0xFFFFFFFFu) have = 0; else if (have) sum_dist += prob * (double)d;
        if (distances_out) distances_out[i] = d;
    }
    *weighted_dist = have ? sum_dist / sum_prob : NAN;
}

/* Test hook: make the solver consume externally supplied tables (the ones the GPU built), so that oracle and GPU
 * chains see bit-identical inputs and their trajectories can be compared exactly. */
void orc_locus_inject_depth_table(orc_locus* l, uint32_t width, const double* table) {
    /* values of BayesCalc::ln_pmf (bayes.rs:27-35) for depths the reference evaluates on the fly: [101][width]; width = 0 removes them */
    free(l->depth_ext_inj); l->depth_ext_inj = NULL; l->depth_ext_width = 0;
    if (width && table) {
        l->depth_ext_inj = (double*)malloc(sizeof(double) * LCTY_GC_BINS * (size_t)width);
        memcpy(l->depth_ext_inj, table, sizeof(double) * LCTY_GC_BINS * (size_t)width);
        l->depth_ext_width = width;
    }
}
void orc_locus_inject_tables(orc_locus* l, const double* depth_lut, const double* win_weight) {
    if (depth_lut) memcpy(l->depth_lut, depth_lut, sizeof(double) * LCTY_GC_BINS * LCTY_DEPTH_CACHE);
    if (win_weight) {
        uint64_t total = 0;
        free(l->ci_off_inj); free(l->win_weight_inj);
        l->ci_off_inj = (uint64_t*)malloc(sizeof(uint64_t) * (l->n_alleles + 1));
        for (uint32_t a = 0; a < l->n_alleles; a++) { l->ci_off_inj[a] = total; total += l->infos[a].n_pos; }
        l->ci_off_inj[l->n_alleles] = total;
        l->win_weight_inj = (double*)malloc(sizeof(double) * (total ? total : 1));
        memcpy(l->win_weight_inj, win_weight, sizeof(double) * total);
    }
}
