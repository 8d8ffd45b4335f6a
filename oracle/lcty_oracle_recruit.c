/*
 * lcty_oracle_recruit.c — CPU restatement of minimizer read recruitment: the step of
 * `locityper genotype` immediately before the hot path (src/seq/recruit.rs, src/seq/kmers.rs:71-340, src/math/frac.rs).
 * TEST INFRASTRUCTURE ONLY — see lcty_oracle.h. PARITY UNPINNED (the reference has no tests); pinned instead by known answers
 * and by an independent Python transliteration (tests/pyref_recruit.py).
 */
#include "lcty_oracle_internal.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MIN(a, b) ((a) < (b) ? (a) : (b))
#define MAX(a, b) ((a) > (b) ? (a) : (b))
#define UNDEF64 0xFFFFFFFFFFFFFFFFull
#define MAX_W 64u                       /* kmers.rs:204 */
#define SUBSUM_BONUS 3u                 /* recruit.rs:40-41 */
#define SUBSUM_PENALTY 1u
#define READ_LENGTH_THRESH 500u         /* recruit.rs:35 */

/* Minimizer for u64: fasthash mix — kmers.rs:93-103 */
uint64_t orc_fast_hash64(uint64_t x) {
    x = ~x;
    x ^= x >> 23;
    x *= 0x2127599bf4325c37ull;
    x ^= x >> 47;
    return x;
}

/* minimizers::<u64, _, CANONICAL> — kmers.rs:265-331. Emits (position, hash, forward); returns the count. */
size_t orc_canon_minimizers(const uint8_t* seq, size_t n, uint32_t k, uint32_t w, uint32_t* pos_out, uint64_t* hash_out, uint8_t* fw_out, size_t cap) {
    const uint64_t mask = (1ull << (2 * k)) - 1;
    const uint32_t rv_shift = 2 * k - 2;
    uint64_t fw_kmer = 0, rv_kmer = 0;
    const uint32_t k_1 = k - 1, w_1 = w - 1;
    uint64_t hashes[MAX_W]; uint8_t forward[MAX_W];
    for (uint32_t i = 0; i < MAX_W; i++) { hashes[i] = UNDEF64; forward[i] = 1; }
    int64_t last_pos = -1;
    uint32_t best_pos = 0;
    uint64_t best_hash = UNDEF64;
    uint32_t first_kmer = k_1, first_window = k_1 + w_1;
    size_t n_out = 0;
    for (size_t ii = 0; ii < n; ii++) {
        const uint32_t i = (uint32_t)ii;
        uint64_t fw_enc, rv_enc;
        switch (seq[ii]) {
            case 'A': fw_enc = 0; rv_enc = 3; break;
            case 'C': fw_enc = 1; rv_enc = 2; break;
            case 'G': fw_enc = 2; rv_enc = 1; break;
            case 'T': fw_enc = 3; rv_enc = 0; break;
            default: first_kmer = i + k; fw_enc = 0; rv_enc = 0;
        }
        fw_kmer = ((fw_kmer << 2) | fw_enc) & mask;
        rv_kmer = (rv_kmer >> 2) | (rv_enc << rv_shift);
        uint64_t kmer; uint8_t fw;
        if (rv_kmer < fw_kmer) { kmer = rv_kmer; fw = 0; } else { kmer = fw_kmer; fw = 1; }
        const uint64_t h = i < first_kmer ? UNDEF64 : orc_fast_hash64(kmer);
        hashes[i & (MAX_W - 1)] = h; forward[i & (MAX_W - 1)] = fw;
        if (h < best_hash) { best_hash = h; best_pos = i; }
        if (i < first_window) continue;
        const uint32_t start = i - w_1;
        if (best_pos < start) {
            best_pos = start; best_hash = hashes[start & (MAX_W - 1)];           /* find_min, kmers.rs:243-258 */
            for (uint32_t j = start + 1; j < i + 1; j++) if (hashes[j & (MAX_W - 1)] < best_hash) { best_pos = j; best_hash = hashes[j & (MAX_W - 1)]; }
            if (best_hash == UNDEF64) { first_window = first_window + w_1; continue; }
        }
        if ((int64_t)best_pos > last_pos) {
            last_pos = best_pos;
            if (n_out < cap) {
                if (pos_out) pos_out[n_out] = best_pos - k_1;
                hash_out[n_out] = best_hash; fw_out[n_out] = forward[best_pos & (MAX_W - 1)];
            }
            n_out++;
        }
    }
    return n_out;
}

/* Fraction::<u16>::approximate — math/frac.rs:50-76 */
void orc_fraction_approximate_u16(double x, uint16_t* num, uint16_t* den) {
    uint32_t a2 = 1, a1 = (uint32_t)floor(x), b2 = 0, b1 = 1;
    double xk = x;
    for (int it = 0; it < 20; it++) {
        const double numer = xk - floor(xk);
        if (numer <= 2.220446049250313e-16) break;
        xk = 1.0 / numer;
        const double fl = floor(xk);
        if (!(fl >= 0.0 && fl <= 65535.0)) break;                          /* T::from_f64 */
        const uint64_t f = (uint64_t)fl;
        const uint64_t a0 = f * a1 + a2, b0 = f * b1 + b2;
        if (f * a1 > 65535 || a0 > 65535 || f * b1 > 65535 || b0 > 65535) break;   /* checked_mul / checked_add */
        a2 = a1; a1 = (uint32_t)a0; b2 = b1; b1 = (uint32_t)b0;
        if (fabs((double)a1 / (double)b1 - x) <= 2.220446049250313e-16) break;
    }
    *num = (uint16_t)a1; *den = (uint16_t)b1;
}

/* ---- targets ---- */
typedef struct { uint64_t minim; uint32_t locus; uint8_t direction, rare; } tentry;     /* MinimInfo: direction bit 0 backward, bit 1 forward */
struct orc_targets {
    uint8_t k, w;
    double match_frac; uint16_t mf_num, mf_den;
    uint32_t match_length, stretch_minims, stretch_score;
    uint16_t thresh_kmer_count;
    uint32_t n_loci;
    tentry* e; size_t n, cap;        /* one entry per (minimizer, locus); sorted by (minim, locus) after finalize */
    int sorted;
};

/* Params::new — recruit.rs:65-105 */
orc_targets* orc_targets_new(uint8_t k, uint8_t w, double match_frac, uint32_t match_length, uint16_t thresh_kmer_count) {
    orc_targets* t = (orc_targets*)calloc(1, sizeof(orc_targets));
    t->k = k; t->w = w; t->match_frac = match_frac; t->match_length = match_length; t->thresh_kmer_count = thresh_kmer_count;
    t->stretch_minims = (2 * match_length + ((uint32_t)w + 1) - 1) / ((uint32_t)w + 1);         /* fast_ceil_div */
    double score = (double)t->stretch_minims * ((double)(SUBSUM_BONUS + SUBSUM_PENALTY) * match_frac - (double)SUBSUM_PENALTY);
    score = ceil(fmax(score, (double)SUBSUM_BONUS));
    t->stretch_score = (uint32_t)score;
    orc_fraction_approximate_u16(match_frac, &t->mf_num, &t->mf_den);
    return t;
}
void orc_targets_free(orc_targets* t) { if (t) { free(t->e); free(t); } }
void orc_targets_params(const orc_targets* t, uint16_t* mf_num, uint16_t* mf_den, uint32_t* stretch_minims, uint32_t* stretch_score) {
    *mf_num = t->mf_num; *mf_den = t->mf_den; *stretch_minims = t->stretch_minims; *stretch_score = t->stretch_score;
}

static int cmp_tentry(const void* x, const void* y) {
    const tentry* a = (const tentry*)x; const tentry* b = (const tentry*)y;
    if (a->minim != b->minim) return a->minim < b->minim ? -1 : 1;
    return a->locus < b->locus ? -1 : (a->locus > b->locus ? 1 : 0);
}

/* TargetBuilder::add — recruit.rs:688-738 (one call = one locus) */
uint32_t orc_targets_add_locus(orc_targets* t, uint32_t n_alleles, const uint8_t* seqs, const uint64_t* seq_off, const uint16_t* counts,
                               const uint64_t* cnt_off, uint32_t base_k) {
    const uint32_t locus = t->n_loci++;
    const uint32_t mk = t->k;
    const size_t shift = mk <= base_k ? (base_k - mk) / 2 : mk - base_k;
    size_t first = t->n;
    for (uint32_t a = 0; a < n_alleles; a++) {
        const uint8_t* seq = seqs + seq_off[a];
        const size_t len = (size_t)(seq_off[a + 1] - seq_off[a]);
        const uint16_t* cnt = counts + cnt_off[a];
        const size_t n_counts = (size_t)(cnt_off[a + 1] - cnt_off[a]);
        size_t cap = len + 1;
        uint32_t* pos = (uint32_t*)malloc(sizeof(uint32_t) * cap); uint64_t* hs = (uint64_t*)malloc(sizeof(uint64_t) * cap); uint8_t* fw = (uint8_t*)malloc(cap);
        const size_t nm = orc_canon_minimizers(seq, len, mk, t->w, pos, hs, fw, cap);
        for (size_t i = 0; i < nm; i++) {
            const size_t p = pos[i];
            int rare;
            if (mk <= base_k) {
                const size_t q = p > shift ? p - shift : 0;
                rare = cnt[MIN(q, n_counts - 1)] < t->thresh_kmer_count;
            } else rare = cnt[p] < t->thresh_kmer_count && cnt[p + shift] < t->thresh_kmer_count;
            if (t->n == t->cap) { t->cap = t->cap ? 2 * t->cap : 1024; t->e = (tentry*)realloc(t->e, sizeof(tentry) * t->cap); }
            tentry* e = &t->e[t->n++];
            e->minim = hs[i]; e->locus = locus; e->direction = (uint8_t)(1 + fw[i]); e->rare = (uint8_t)rare;
        }
        free(pos); free(hs); free(fw);
    }
    /* MinimInfo::update within the locus: directions OR-ed, rare AND-ed */
    qsort(t->e + first, t->n - first, sizeof(tentry), cmp_tentry);
    size_t wr = first;
    for (size_t i = first; i < t->n; i++) {
        if (wr > first && t->e[wr - 1].minim == t->e[i].minim) { t->e[wr - 1].direction |= t->e[i].direction; t->e[wr - 1].rare &= t->e[i].rare; }
        else t->e[wr++] = t->e[i];
    }
    t->n = wr; t->sorted = 0;
    return locus;
}
void orc_targets_finalize(orc_targets* t) { qsort(t->e, t->n, sizeof(tentry), cmp_tentry); t->sorted = 1; }
size_t orc_targets_n_entries(const orc_targets* t) { return t->n; }
void orc_targets_entry(const orc_targets* t, size_t i, uint64_t* minim, uint32_t* locus, uint8_t* direction, uint8_t* rare) {
    *minim = t->e[i].minim; *locus = t->e[i].locus; *direction = t->e[i].direction; *rare = t->e[i].rare;
}

static size_t lower_bound(const orc_targets* t, uint64_t minim) {
    size_t lo = 0, hi = t->n;
    while (lo < hi) { const size_t mid = (lo + hi) / 2; if (t->e[mid].minim < minim) lo = mid + 1; else hi = mid; }
    return lo;
}

/* BaseMatchCount: [common-backward, common-forward, rare-backward, rare-forward] — recruit.rs:236-262 */
typedef struct { uint32_t locus; uint32_t first[4], second[4]; } mcount;
static inline void mc_inc(uint32_t* arr, int forward, const tentry* e) {
    const uint32_t i = (uint32_t)e->rare << 1;
    arr[i] += (e->direction & (1u + (uint32_t)!forward)) != 0;
    arr[i | 1] += (e->direction & (1u + (uint32_t)forward)) != 0;
}
static inline int has_rare(const uint32_t* a) { return a[2] != 0 || a[3] != 0; }
#define WORTH 3u
static inline uint16_t fw_num(const uint32_t* a) { return (uint16_t)(WORTH * a[3] + a[1]); }
static inline uint16_t bw_num(const uint32_t* a) { return (uint16_t)(WORTH * a[2] + a[0]); }
static inline uint16_t fw_den(const uint32_t* a, uint32_t total) { return (uint16_t)(WORTH * (total - a[1]) + a[1]); }
static inline uint16_t bw_den(const uint32_t* a, uint32_t total) { return (uint16_t)(WORTH * (total - a[0]) + a[0]); }
static inline int frac_ge(uint16_t n1, uint16_t d1, uint16_t n2, uint16_t d2) { return (uint32_t)n1 * d2 >= (uint32_t)n2 * d1; }   /* frac.rs:87-93 */

typedef struct { mcount* v; size_t n, cap; } matches;
static mcount* m_get(matches* m, uint32_t locus, int insert) {
    for (size_t i = 0; i < m->n; i++) if (m->v[i].locus == locus) return &m->v[i];
    if (!insert) return NULL;
    if (m->n == m->cap) { m->cap = m->cap ? 2 * m->cap : 8; m->v = (mcount*)realloc(m->v, sizeof(mcount) * m->cap); }
    mcount* c = &m->v[m->n++]; memset(c, 0, sizeof(*c)); c->locus = locus;
    return c;
}
static int cmp_u32(const void* a, const void* b) { const uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b; return x < y ? -1 : (x > y ? 1 : 0); }

/* has_matching_stretch — recruit.rs:938-961 */
static int has_matching_stretch(const orc_targets* t, uint32_t locus, const uint64_t* hs, const uint8_t* fw, size_t n) {
    uint32_t s_fw = 0, s_bw = 0;
    for (size_t i = 0; i < n; i++) {
        for (size_t j = lower_bound(t, hs[i]); j < t->n && t->e[j].minim == hs[i]; j++) if (t->e[j].locus == locus) {
            const uint32_t x = SUBSUM_PENALTY + (uint32_t)t->e[j].rare * SUBSUM_BONUS;
            s_fw += ((t->e[j].direction & (1u + (uint32_t)fw[i])) != 0) * x;
            s_bw += ((t->e[j].direction & (1u + (uint32_t)!fw[i])) != 0) * x;
        }
        s_fw = s_fw > SUBSUM_PENALTY ? s_fw - SUBSUM_PENALTY : 0;
        s_bw = s_bw > SUBSUM_PENALTY ? s_bw - SUBSUM_PENALTY : 0;
        if (s_fw >= t->stretch_score || s_bw >= t->stretch_score) return 1;
    }
    return 0;
}

/* Targets::recruit_read_pair (recruit.rs:883-929) when seq2 != NULL; otherwise recruit_short_read (848-879) for reads of up to
 * 500 bases and recruit_long_read (964-996) beyond (the dispatch of 589-595). Writes the loci in increasing order; returns their number. */
size_t orc_recruit(const orc_targets* t, const uint8_t* seq1, size_t n1, const uint8_t* seq2, size_t n2, uint32_t* out, size_t cap) {
    size_t c1 = n1 + 1, c2 = n2 + 1;
    uint64_t* h1 = (uint64_t*)malloc(sizeof(uint64_t) * c1); uint8_t* f1 = (uint8_t*)malloc(c1);
    uint64_t* h2 = (uint64_t*)malloc(sizeof(uint64_t) * c2); uint8_t* f2 = (uint8_t*)malloc(c2);
    matches m = {0};
    size_t n_out = 0;
    const size_t total1 = orc_canon_minimizers(seq1, n1, t->k, t->w, NULL, h1, f1, c1);
    for (size_t i = 0; i < total1; i++)
        for (size_t j = lower_bound(t, h1[i]); j < t->n && t->e[j].minim == h1[i]; j++) mc_inc(m_get(&m, t->e[j].locus, 1)->first, f1[i], &t->e[j]);
    uint32_t* ans = (uint32_t*)malloc(sizeof(uint32_t) * (m.n ? m.n : 1));
    if (seq2) {
        if (m.n) {
            const size_t total2 = orc_canon_minimizers(seq2, n2, t->k, t->w, NULL, h2, f2, c2);
            for (size_t i = 0; i < total2; i++)
                for (size_t j = lower_bound(t, h2[i]); j < t->n && t->e[j].minim == h2[i]; j++) {
                    mcount* c = m_get(&m, t->e[j].locus, 0);
                    if (c) mc_inc(c->second, f2[i], &t->e[j]);
                }
            for (size_t q = 0; q < m.n; q++) {
                const mcount* c = &m.v[q];
                if (!(has_rare(c->first) || has_rare(c->second))) continue;
                uint16_t na, da, nb, db;                                     /* better_pair_fraction, recruit.rs:351-367 */
                if ((uint16_t)(fw_num(c->first) + bw_num(c->second)) >= (uint16_t)(bw_num(c->first) + fw_num(c->second))) {
                    na = fw_num(c->first); da = fw_den(c->first, (uint32_t)total1); nb = bw_num(c->second); db = bw_den(c->second, (uint32_t)total2);
                } else {
                    na = bw_num(c->first); da = bw_den(c->first, (uint32_t)total1); nb = fw_num(c->second); db = fw_den(c->second, (uint32_t)total2);
                }
                if (frac_ge(na, da, t->mf_num, t->mf_den) && frac_ge(nb, db, t->mf_num, t->mf_den)) ans[n_out++] = c->locus;
            }
        }
    } else if (n1 <= READ_LENGTH_THRESH) {
        for (size_t q = 0; q < m.n; q++) {
            const mcount* c = &m.v[q];
            if (!has_rare(c->first)) continue;
            uint16_t na, da;                                                 /* better_fraction, recruit.rs:340-348 */
            if (fw_num(c->first) >= bw_num(c->first)) { na = fw_num(c->first); da = fw_den(c->first, (uint32_t)total1); }
            else { na = bw_num(c->first); da = bw_den(c->first, (uint32_t)total1); }
            if (frac_ge(na, da, t->mf_num, t->mf_den)) ans[n_out++] = c->locus;
        }
    } else {
        for (size_t q = 0; q < m.n; q++) {
            const mcount* c = &m.v[q];
            const uint32_t* a = c->first;                                    /* rare_fraction, recruit.rs:271-279 */
            uint32_t num, den;
            if (a[3] >= a[2]) { num = a[3]; den = (uint32_t)total1 - a[1]; } else { num = a[2]; den = (uint32_t)total1 - a[0]; }
            const uint32_t thr = MAX(1u, (uint32_t)ceil((double)MIN(t->stretch_minims, den) * t->match_frac));       /* long_read_threshold */
            if (num >= thr && (den < t->stretch_minims || has_matching_stretch(t, c->locus, h1, f1, total1))) ans[n_out++] = c->locus;
        }
    }
    qsort(ans, n_out, sizeof(uint32_t), cmp_u32);
    for (size_t i = 0; i < n_out && i < cap; i++) out[i] = ans[i];
    free(ans); free(m.v); free(h1); free(f1); free(h2); free(f2);
    return n_out;
}
