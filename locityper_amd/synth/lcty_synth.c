/*
 * lcty_synth.c — deterministic synthetic workloads of the shapes named in BASELINE.json
 * (SURVEY.md §8d): alleles on a coalescent-like tree, off-target k-mer counts, a BgDistr,
 * read pairs and their candidate alignments in the input-order contract of
 * include/locityper_hip.h (lcty_reads_host).
 *
 * Bench / test input generator. Not part of the product path and not part of the oracle:
 * both consume exactly the buffers produced here.
 *
 * Every read pair is generated from (seed, pair index) alone, so any sub-range of the
 * workload can be produced independently (chunked upload, CPU-baseline sample) and the
 * result does not depend on the number of OpenMP threads.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "../../include/locityper_hip.h"

#define MIN(a, b) ((a) < (b) ? (a) : (b))
#define MAX(a, b) ((a) > (b) ? (a) : (b))

/* ---- RNG: SplitMix64 seeding + xoshiro256++ (public-domain constants) ---- */
typedef struct { uint64_t s[4]; } rng_t;
static inline uint64_t splitmix64(uint64_t* x) {
    uint64_t z = (*x += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static inline uint64_t hash2(uint64_t a, uint64_t b) {
    uint64_t x = a ^ (b * 0xd6e8feb86659fd93ULL + 0x2545F4914F6CDD1DULL);
    return splitmix64(&x);
}
static inline void rng_seed(rng_t* r, uint64_t seed) {
    for (int i = 0; i < 4; i++) r->s[i] = splitmix64(&seed);
}
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t rng_next(rng_t* r) {
    uint64_t* s = r->s;
    uint64_t result = rotl(s[0] + s[3], 23) + s[0];
    uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return result;
}
static inline uint32_t rng_below(rng_t* r, uint32_t n) {   /* mulhi range reduction */
    return (uint32_t)(((rng_next(r) >> 32) * (uint64_t)n) >> 32);
}
static inline double rng_unif(rng_t* r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }
static inline double rng_gauss(rng_t* r) {
    double u1 = rng_unif(r), u2 = rng_unif(r);
    if (u1 < 1e-300) u1 = 1e-300;
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

/* ---- locus ---------------------------------------------------------------- */
typedef struct {
    uint32_t pos;       /* base-haplotype coordinate */
    uint8_t type;       /* 0 SNP, 1 insertion (allele has extra bases after pos), 2 deletion (allele lacks [pos, pos+len)) */
    uint8_t alt;        /* SNP: alternative base code */
    uint16_t len;
    uint32_t node;      /* clade: carried by alleles with (leaf >> (depth_max - depth)) == node */
    uint8_t depth;
    uint64_t ins_seed;
} variant;

typedef struct { uint32_t base_pos; int32_t shift_after; } coord_bp;   /* allele = base + shift for base >= base_pos */

typedef struct synth_locus {
    uint64_t seed;
    uint32_t n_alleles, base_len, k, depth_max;
    int technology;
    uint32_t read_len;
    uint8_t* base;                 /* ASCII */
    variant* vars; uint32_t n_vars;   /* sorted by pos */
    uint32_t* leaf;                /* allele -> leaf index of the clade tree */
    uint8_t* seqs; uint64_t* seq_off;
    uint16_t* counts; uint64_t* cnt_off;
    coord_bp** bps; uint32_t* n_bps;  /* per allele, sorted by base_pos */
    lcty_bg bg;
    uint32_t true_gt[2];
    uint64_t n_pairs_expected;
} synth_locus;

static inline int carries(const synth_locus* L, uint32_t a, const variant* v) {
    return (L->leaf[a] >> (L->depth_max - v->depth)) == v->node;
}

/* base -> allele coordinate (position of the allele base that descends from base position p;
 * inside a deletion the next surviving base) */
static uint32_t b2a(const synth_locus* L, uint32_t a, uint32_t p) {
    const coord_bp* bp = L->bps[a]; uint32_t n = L->n_bps[a];
    uint32_t lo = 0, hi = n;   /* last breakpoint with base_pos <= p */
    while (lo < hi) { uint32_t mid = (lo + hi) / 2; if (bp[mid].base_pos <= p) lo = mid + 1; else hi = mid; }
    int32_t shift = lo ? bp[lo - 1].shift_after : 0;
    int64_t r = (int64_t)p + shift;
    return r < 0 ? 0 : (uint32_t)r;
}
/* allele -> base coordinate (approximate inverse of b2a) */
static uint32_t a2b(const synth_locus* L, uint32_t a, uint32_t q) {
    const coord_bp* bp = L->bps[a]; uint32_t n = L->n_bps[a];
    int32_t shift = 0;
    for (uint32_t lo = 0, hi = n; lo < hi;) {
        uint32_t mid = (lo + hi) / 2;
        int64_t apos = (int64_t)bp[mid].base_pos + bp[mid].shift_after;
        if (apos <= (int64_t)q) { shift = bp[mid].shift_after; lo = mid + 1; } else hi = mid;
    }
    int64_t r = (int64_t)q - shift;
    if (r < 0) r = 0;
    if (r >= (int64_t)L->base_len) r = L->base_len - 1;
    return (uint32_t)r;
}

static int cmp_var(const void* x, const void* y) {
    const variant* a = (const variant*)x; const variant* b = (const variant*)y;
    return a->pos < b->pos ? -1 : (a->pos > b->pos ? 1 : 0);
}

void synth_locus_free(synth_locus* L) {
    if (!L) return;
    free(L->base); free(L->vars); free(L->leaf); free(L->seqs); free(L->seq_off);
    free(L->counts); free(L->cnt_off);
    if (L->bps) for (uint32_t a = 0; a < L->n_alleles; a++) free(L->bps[a]);
    free(L->bps); free(L->n_bps);
    free(L);
}

/* technology: LCTY_TECH_ILLUMINA (150 bp PE) or LCTY_TECH_NANOPORE (long SE reads, read_len mean) */
synth_locus* synth_locus_new(uint64_t seed, uint32_t n_alleles, uint32_t base_len, uint32_t k,
                             int technology, uint32_t read_len, uint64_t n_pairs_expected) {
    synth_locus* L = (synth_locus*)calloc(1, sizeof(synth_locus));
    L->seed = seed; L->n_alleles = n_alleles; L->base_len = base_len; L->k = k;
    L->technology = technology; L->read_len = read_len; L->n_pairs_expected = n_pairs_expected;
    rng_t r; rng_seed(&r, hash2(seed, 0x10c05));

    /* base haplotype with 20 planted 500-bp low-complexity blocks */
    L->base = (uint8_t*)malloc(base_len);
    for (uint32_t i = 0; i < base_len; i++) L->base[i] = "ACGT"[rng_below(&r, 4)];
    uint8_t* lowc = (uint8_t*)calloc(base_len, 1);
    uint32_t n_blocks = base_len >= 20000 ? 20 : base_len / 2500;
    for (uint32_t b = 0; b < n_blocks; b++) {
        uint32_t blen = 500, start = 300 + rng_below(&r, base_len - blen - 600);
        uint32_t mlen = 1 + rng_below(&r, 6);
        uint8_t motif[6];
        for (uint32_t j = 0; j < mlen; j++) motif[j] = "ACGT"[rng_below(&r, 4)];
        for (uint32_t j = 0; j < blen; j++) { L->base[start + j] = motif[j % mlen]; lowc[start + j] = 1; }
    }

    /* clade tree: leaves = random permutation of alleles over 2^depth_max slots */
    uint32_t D = 0; while ((1u << D) < n_alleles) D++;
    L->depth_max = D;
    L->leaf = (uint32_t*)malloc(sizeof(uint32_t) * n_alleles);
    {
        uint32_t nl = 1u << D;
        uint32_t* perm = (uint32_t*)malloc(sizeof(uint32_t) * nl);
        for (uint32_t i = 0; i < nl; i++) perm[i] = i;
        for (uint32_t i = nl - 1; i > 0; i--) { uint32_t j = rng_below(&r, i + 1); uint32_t t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
        for (uint32_t a = 0; a < n_alleles; a++) L->leaf[a] = perm[a];
        free(perm);
    }

    /* variant sites: SNPs 0.5 %, indels 0.05 % per allele relative to the base */
    double carrier = 0.0; for (uint32_t d = 0; d <= D; d++) carrier += pow(0.5, (double)d);
    carrier /= (double)(D + 1);
    uint32_t n_snp = (uint32_t)((double)base_len * 0.005 / carrier);
    uint32_t n_indel = (uint32_t)((double)base_len * 0.0005 / carrier);
    uint32_t n_vars = n_snp + n_indel;
    variant* vars = (variant*)calloc(n_vars ? n_vars : 1, sizeof(variant));
    for (uint32_t i = 0; i < n_vars; i++) {
        variant* v = &vars[i];
        v->pos = 100 + rng_below(&r, base_len - 200);
        v->depth = (uint8_t)rng_below(&r, D + 1);
        v->node = rng_below(&r, 1u << v->depth);
        if (i < n_snp) {
            v->type = 0; v->len = 1;
            v->alt = (uint8_t)rng_below(&r, 3);     /* resolved against the base below */
        } else {
            v->type = (uint8_t)(1 + rng_below(&r, 2));
            uint32_t len = 1; while (len < 30 && rng_unif(&r) < 0.7) len++;   /* geometric, 1..30 */
            v->len = (uint16_t)len;
            v->ins_seed = rng_next(&r);
        }
    }
    qsort(vars, n_vars, sizeof(variant), cmp_var);
    /* keep sites apart: drop any site closer than (len + 2) to the previous kept one */
    uint32_t w = 0, next_free = 0;
    for (uint32_t i = 0; i < n_vars; i++) {
        if (vars[i].pos < next_free) continue;
        vars[w] = vars[i];
        next_free = vars[i].pos + (vars[i].type == 2 ? vars[i].len : 1) + 2;
        w++;
    }
    n_vars = w;
    for (uint32_t i = 0; i < n_vars; i++) if (vars[i].type == 0) {
        uint8_t b = L->base[vars[i].pos];
        int bc = b == 'A' ? 0 : b == 'C' ? 1 : b == 'G' ? 2 : 3;
        vars[i].alt = (uint8_t)((bc + 1 + vars[i].alt) & 3);
    }
    L->vars = vars; L->n_vars = n_vars;

    /* allele sequences + coordinate maps */
    L->seq_off = (uint64_t*)calloc(n_alleles + 1, sizeof(uint64_t));
    L->bps = (coord_bp**)calloc(n_alleles, sizeof(coord_bp*));
    L->n_bps = (uint32_t*)calloc(n_alleles, sizeof(uint32_t));
    uint64_t cap = (uint64_t)n_alleles * ((uint64_t)base_len + 64 * 32) + 1024;
    L->seqs = (uint8_t*)malloc(cap);
    uint64_t off = 0;
    for (uint32_t a = 0; a < n_alleles; a++) {
        L->seq_off[a] = off;
        coord_bp* bp = (coord_bp*)malloc(sizeof(coord_bp) * (n_vars + 1));
        uint32_t nb = 0; int32_t shift = 0;
        uint32_t p = 0;
        for (uint32_t i = 0; i < n_vars; i++) {
            const variant* v = &vars[i];
            if (!carries(L, a, v)) continue;
            if (v->pos < p) continue;        /* swallowed by a previous deletion */
            memcpy(L->seqs + off, L->base + p, v->pos - p); off += v->pos - p; p = v->pos;
            if (v->type == 0) {
                L->seqs[off++] = "ACGT"[v->alt]; p++;
            } else if (v->type == 1) {
                L->seqs[off++] = L->base[p]; p++;
                rng_t ir; rng_seed(&ir, v->ins_seed);
                for (uint32_t j = 0; j < v->len; j++) L->seqs[off++] = "ACGT"[rng_below(&ir, 4)];
                shift += v->len;
                bp[nb].base_pos = p; bp[nb].shift_after = shift; nb++;
            } else {
                p += v->len;
                shift -= v->len;
                bp[nb].base_pos = p; bp[nb].shift_after = shift; nb++;
            }
        }
        memcpy(L->seqs + off, L->base + p, base_len - p); off += base_len - p;
        L->bps[a] = (coord_bp*)realloc(bp, sizeof(coord_bp) * (nb ? nb : 1));
        L->n_bps[a] = nb;
    }
    L->seq_off[n_alleles] = off;

    /* off-target counts: 0 for ~90 % of base positions, runs (50-500) of 1..200 elsewhere;
     * the planted low-complexity blocks are always off-target */
    uint16_t* base_cnt = (uint16_t*)calloc(base_len, sizeof(uint16_t));
    {
        uint64_t covered = 0, target = base_len / 10;
        while (covered < target) {
            uint32_t rl = 50 + rng_below(&r, 451), st = rng_below(&r, base_len - rl);
            uint16_t val = (uint16_t)(1 + rng_below(&r, 200));
            for (uint32_t j = 0; j < rl; j++) base_cnt[st + j] = val;
            covered += rl;
        }
        for (uint32_t i = 0; i < base_len; i++) if (lowc[i] && base_cnt[i] == 0) base_cnt[i] = 50;
    }
    L->cnt_off = (uint64_t*)calloc(n_alleles + 1, sizeof(uint64_t));
    uint64_t tot_cnt = 0;
    for (uint32_t a = 0; a < n_alleles; a++) {
        L->cnt_off[a] = tot_cnt;
        tot_cnt += (L->seq_off[a + 1] - L->seq_off[a]) + 1 - k;
    }
    L->cnt_off[n_alleles] = tot_cnt;
    L->counts = (uint16_t*)malloc(sizeof(uint16_t) * (tot_cnt ? tot_cnt : 1));
    for (uint32_t a = 0; a < n_alleles; a++) {
        uint64_t n = L->cnt_off[a + 1] - L->cnt_off[a];
        for (uint64_t i = 0; i < n; i++) L->counts[L->cnt_off[a] + i] = base_cnt[a2b(L, a, (uint32_t)i)];
    }
    free(base_cnt); free(lowc);

    /* BgDistr */
    lcty_bg* bg = &L->bg; memset(bg, 0, sizeof(*bg));
    int ont = technology != LCTY_TECH_ILLUMINA;
    double pm, px, pi_, pd;
    if (!ont) { px = 0.003; pi_ = 0.001; pd = 0.001; } else { px = 0.01; pi_ = 0.01; pd = 0.01; }
    pm = 1.0 - px - pi_ - pd;
    bg->op_lnprobs[0] = log(pm); bg->op_lnprobs[1] = log(px); bg->op_lnprobs[2] = log(pi_);
    bg->op_lnprobs[3] = log(pd); bg->op_lnprobs[4] = log(MAX(px, pi_));
    bg->edit_alpha = ont ? 6.0 : 0.6; bg->edit_beta = ont ? 180.0 : 90.0;
    bg->is_paired = !ont;
    {   /* NBinom::estimate_corrected(mean 450, var 80^2) — nbinom.rs:53-65 */
        double m = 450.0, v = 6400.0;
        bg->ins_n = m * m / (v - m); bg->ins_p = m / v;
    }
    bg->window = ont ? MIN(5000u, MAX(20u, (uint32_t)round(2.0 / 3.0 * read_len))) : 100;
    bg->neighb = MAX(300u, bg->window);
    {
        double per_hap = (double)n_pairs_expected / 2.0;
        double m0 = per_hap * (double)bg->window / (double)base_len;
        if (m0 < 0.5) m0 = 0.5;
        for (int gc = 0; gc < LCTY_GC_BINS; gc++) {
            double m = m0 * (0.85 + 0.3 * (double)gc / 100.0), v = 1.5 * m;
            bg->depth_n[gc] = m * m / (v - m); bg->depth_p[gc] = m / v;
        }
    }
    bg->technology = technology;
    if (!ont) { bg->edit_kind = LCTY_EDIT_FRACTION; bg->edit_p1 = 0.03; bg->edit_p2 = 0.06; }
    else { bg->edit_kind = LCTY_EDIT_PVALUE; bg->edit_p1 = 0.99; bg->edit_p2 = 0.999; }

    L->true_gt[0] = rng_below(&r, n_alleles);
    do { L->true_gt[1] = rng_below(&r, n_alleles); } while (n_alleles > 1 && L->true_gt[1] == L->true_gt[0]);
    if (L->true_gt[0] > L->true_gt[1]) { uint32_t t = L->true_gt[0]; L->true_gt[0] = L->true_gt[1]; L->true_gt[1] = t; }
    return L;
}

uint32_t synth_locus_n_alleles(const synth_locus* L) { return L->n_alleles; }
uint32_t synth_locus_k(const synth_locus* L) { return L->k; }
const uint8_t* synth_locus_seqs(const synth_locus* L) { return L->seqs; }
const uint64_t* synth_locus_seq_off(const synth_locus* L) { return L->seq_off; }
const uint16_t* synth_locus_counts(const synth_locus* L) { return L->counts; }
const uint64_t* synth_locus_cnt_off(const synth_locus* L) { return L->cnt_off; }
const lcty_bg* synth_locus_bg(const synth_locus* L) { return &L->bg; }
void synth_locus_true_genotype(const synth_locus* L, uint32_t* out) { out[0] = L->true_gt[0]; out[1] = L->true_gt[1]; }

/* ---- reads ------------------------------------------------------------------ */
typedef struct { uint32_t q; uint8_t op; uint16_t len; } event;   /* op: 1 X, 2 I, 3 D */
#define MAX_EVENTS 4096
#define MAX_CIG 8192

typedef struct {
    uint32_t* cig; uint32_t n;
} cigbuf;

static inline void cig_push(cigbuf* c, uint32_t op, uint32_t len) {
    if (len == 0) return;
    if (c->n && (c->cig[c->n - 1] & 15u) == op) { c->cig[c->n - 1] += len << 4; return; }
    if (c->n < MAX_CIG) c->cig[c->n++] = (len << 4) | op;
}

/* builds the CIGAR of a read of length rl from events sorted by q; returns ref length */
static uint32_t build_cigar(const event* ev, uint32_t n_ev, uint32_t rl, uint32_t clip_l, uint32_t clip_r, cigbuf* out) {
    out->n = 0;
    uint32_t cq = 0, ref = 0;
    uint32_t q_hi = rl - clip_r;
    if (clip_l) { cig_push(out, LCTY_CIGAR_S, clip_l); cq = clip_l; }
    for (uint32_t i = 0; i < n_ev; i++) {
        const event* e = &ev[i];
        uint32_t qcons = e->op == 1 ? 1 : (e->op == 2 ? e->len : 0);
        if (e->q < cq + 1 || e->q + qcons + 1 > q_hi) continue;    /* keep an '=' on both sides */
        cig_push(out, LCTY_CIGAR_EQ, e->q - cq); ref += e->q - cq; cq = e->q;
        if (e->op == 1) { cig_push(out, LCTY_CIGAR_X, 1); cq += 1; ref += 1; }
        else if (e->op == 2) { cig_push(out, LCTY_CIGAR_I, e->len); cq += e->len; }
        else { cig_push(out, LCTY_CIGAR_D, e->len); ref += e->len; }
    }
    cig_push(out, LCTY_CIGAR_EQ, q_hi - cq); ref += q_hi - cq;
    if (clip_r) cig_push(out, LCTY_CIGAR_S, clip_r);
    return ref;
}

static int cmp_event(const void* x, const void* y) {
    const event* a = (const event*)x; const event* b = (const event*)y;
    return a->q < b->q ? -1 : (a->q > b->q ? 1 : 0);
}

typedef struct {
    /* outputs for one pair; NULL pointers = count only */
    lcty_aln_rec* recs; uint32_t* cigar;
    uint64_t n_recs, n_cigar;
} pair_out;

static inline void emit_rec(pair_out* po, uint32_t pos, uint16_t contig, uint16_t flags, const cigbuf* c) {
    if (po->recs) {
        lcty_aln_rec* r = &po->recs[po->n_recs];
        r->pos = pos; r->contig = contig; r->flags = flags; r->n_cigar = c->n; r->cigar_rel = (uint32_t)po->n_cigar;
        memcpy(po->cigar + po->n_cigar, c->cig, sizeof(uint32_t) * c->n);
    }
    po->n_recs++; po->n_cigar += c->n;
}

typedef struct {
    uint32_t len;          /* read length */
    uint32_t h_start;      /* start on the source haplotype */
    int reverse;
    event err[MAX_EVENTS]; uint32_t n_err;
    uint32_t clip_l, clip_r;
    uint8_t* seq;          /* ASCII, len bytes */
} mate_t;

/* Generates one mate: sequence (haplotype orientation) + sequencing-error events */
static void make_mate(const synth_locus* L, rng_t* r, uint32_t h, uint32_t h_start, uint32_t len, int reverse, mate_t* m) {
    const uint8_t* hs = L->seqs + L->seq_off[h];
    uint32_t hl = (uint32_t)(L->seq_off[h + 1] - L->seq_off[h]);
    int ont = L->technology != LCTY_TECH_ILLUMINA;
    double px = ont ? 0.01 : 0.003, pi_ = ont ? 0.01 : 0.001, pd = ont ? 0.01 : 0.001;
    m->len = len; m->h_start = h_start; m->reverse = reverse; m->n_err = 0;
    uint32_t ref = h_start, q = 0;
    while (q < len) {
        double u = rng_unif(r);
        if (ref >= hl) { m->seq[q++] = "ACGT"[rng_below(r, 4)]; continue; }
        if (u < px && q > 0) {
            uint8_t b = hs[ref]; uint8_t nb;
            do { nb = "ACGT"[rng_below(r, 4)]; } while (nb == b);
            m->seq[q] = nb;
            if (m->n_err < MAX_EVENTS) { m->err[m->n_err].q = q; m->err[m->n_err].op = 1; m->err[m->n_err].len = 1; m->n_err++; }
            q++; ref++;
        } else if (u < px + pi_ && q > 0) {
            m->seq[q] = "ACGT"[rng_below(r, 4)];
            if (m->n_err < MAX_EVENTS) { m->err[m->n_err].q = q; m->err[m->n_err].op = 2; m->err[m->n_err].len = 1; m->n_err++; }
            q++;
        } else if (u < px + pi_ + pd && q > 0) {
            if (m->n_err < MAX_EVENTS) { m->err[m->n_err].q = q; m->err[m->n_err].op = 3; m->err[m->n_err].len = 1; m->n_err++; }
            ref++;
        } else {
            m->seq[q++] = hs[ref++];
        }
    }
    m->clip_l = m->clip_r = 0;
    if (rng_unif(r) < 0.03) {
        uint32_t c = 1 + rng_below(r, 8);
        if (c * 4 < len) { if (rng_below(r, 2)) m->clip_l = c; else m->clip_r = c; }
    }
    if (rng_unif(r) < 0.002) m->seq[rng_below(r, len)] = 'N';
}

/* emits the record of mate m (generated from haplotype h) against allele a */
#define MAX_SITES 2048
typedef struct {
    uint32_t bs;                   /* base coordinate of the mate start */
    uint32_t n;
    uint32_t var[MAX_SITES];       /* variant index */
    uint32_t q[MAX_SITES];         /* read offset of the site (non-decreasing) */
    uint8_t ch[MAX_SITES];         /* carried by the source haplotype */
} mate_sites;

/* variant sites under the mate, computed once per mate and reused for every allele */
static void prepare_sites(const synth_locus* L, const mate_t* m, uint32_t h, mate_sites* ms) {
    ms->bs = a2b(L, h, m->h_start);
    ms->n = 0;
    uint32_t be = a2b(L, h, m->h_start + m->len);
    uint32_t lo = 0, hi = L->n_vars;
    while (lo < hi) { uint32_t mid = (lo + hi) / 2; if (L->vars[mid].pos < ms->bs) lo = mid + 1; else hi = mid; }
    for (uint32_t i = lo; i < L->n_vars && L->vars[i].pos < be && ms->n < MAX_SITES; i++) {
        const variant* v = &L->vars[i];
        uint32_t hp = b2a(L, h, v->pos);
        if (hp < m->h_start) continue;
        ms->var[ms->n] = i; ms->q[ms->n] = hp - m->h_start; ms->ch[ms->n] = (uint8_t)carries(L, h, v);
        ms->n++;
    }
}

/* emits the record of mate m (generated from haplotype h) against allele a */
static void align_to(const synth_locus* L, const mate_t* m, const mate_sites* ms, uint32_t h, uint32_t a, uint16_t flags,
                     event* evbuf, cigbuf* cb, pair_out* po) {
    uint32_t n_ev = 0;
    if (a == h) {
        memcpy(evbuf, m->err, sizeof(event) * m->n_err); n_ev = m->n_err;
    } else {
        /* merge the (sorted) sequencing-error events with the (sorted) variant differences */
        uint32_t ie = 0;
        for (uint32_t s = 0; s < ms->n && n_ev < MAX_EVENTS; s++) {
            const variant* v = &L->vars[ms->var[s]];
            int ch = ms->ch[s], ca = carries(L, a, v);
            if (ch == ca) continue;
            event e;
            e.q = ms->q[s];
            if (v->type == 0) { e.op = 1; e.len = 1; }
            else if ((v->type == 1) == (ch != 0)) { e.op = 2; e.len = v->len; e.q += 1; }   /* read has extra bases */
            else { e.op = 3; e.len = v->len; e.q += (v->type == 1); }
            while (ie < m->n_err && m->err[ie].q <= e.q && n_ev < MAX_EVENTS) evbuf[n_ev++] = m->err[ie++];
            if (n_ev < MAX_EVENTS) evbuf[n_ev++] = e;
        }
        while (ie < m->n_err && n_ev < MAX_EVENTS) evbuf[n_ev++] = m->err[ie++];
    }
    uint32_t pos = b2a(L, a, ms->bs) + m->clip_l;
    build_cigar(evbuf, n_ev, m->len, m->clip_l, m->clip_r, cb);
    uint32_t alen = (uint32_t)(L->seq_off[a + 1] - L->seq_off[a]);
    if (pos >= alen) pos = alen - 1;
    emit_rec(po, pos, (uint16_t)a, (uint16_t)(flags | (m->reverse ? LCTY_FLAG_REVERSE : 0)), cb);
}

/* extra record with `n_x` mismatches at `pos` on allele `a` (decoys / near-duplicates) */
static void emit_variant_rec(const synth_locus* L, rng_t* r, const mate_t* m, uint32_t a, uint32_t pos,
                             uint32_t n_x, int reverse, uint16_t flags, event* evbuf, cigbuf* cb, pair_out* po) {
    uint32_t n_ev = 0;
    for (uint32_t i = 0; i < n_x && n_ev < MAX_EVENTS; i++) {
        evbuf[n_ev].q = 2 + rng_below(r, m->len - 4); evbuf[n_ev].op = 1; evbuf[n_ev].len = 1; n_ev++;
    }
    qsort(evbuf, n_ev, sizeof(event), cmp_event);
    build_cigar(evbuf, n_ev, m->len, 0, 0, cb);
    uint32_t alen = (uint32_t)(L->seq_off[a + 1] - L->seq_off[a]);
    if (pos + m->len + 1 >= alen) pos = alen > m->len + 2 ? alen - m->len - 2 : 0;
    emit_rec(po, pos, (uint16_t)a, (uint16_t)(flags | (reverse ? LCTY_FLAG_REVERSE : 0)), cb);
}

typedef struct {
    event* ev; uint32_t* cig; uint8_t* seq1; uint8_t* seq2; mate_t* m1; mate_t* m2; mate_sites* ms;
} scratch;

static void gen_pair(const synth_locus* L, uint64_t pair, scratch* S, pair_out* po, uint32_t* len_out) {
    rng_t r; rng_seed(&r, hash2(L->seed ^ 0x5ca1ab1e, pair));
    cigbuf cb; cb.cig = S->cig; cb.n = 0;
    int paired = L->bg.is_paired;
    uint32_t A = L->n_alleles;
    po->n_recs = 0; po->n_cigar = 0;
    mate_t* m1 = S->m1; mate_t* m2 = S->m2;
    m1->seq = S->seq1; m2->seq = S->seq2;

    uint32_t rl = L->read_len;
    if (!paired) {   /* long reads: lognormal length, sigma 0.2 */
        double f = exp(0.2 * rng_gauss(&r));
        rl = (uint32_t)MAX(500.0, MIN(2.5 * L->read_len, f * L->read_len));
    }
    double u_kind = rng_unif(&r);
    uint32_t h = L->true_gt[rng_below(&r, 2)];
    uint32_t hl = (uint32_t)(L->seq_off[h + 1] - L->seq_off[h]);
    if (rl + 20 > hl) rl = hl - 20;
    uint32_t frag = rl;
    if (paired) {
        double f = 450.0 + 80.0 * rng_gauss(&r);
        frag = (uint32_t)MAX((double)rl + 10.0, MIN(1200.0, f));
        if (frag + 2 > hl) frag = hl - 2;
    }
    uint32_t fstart = rng_below(&r, hl - frag);
    int flip = rng_below(&r, 2);
    len_out[0] = rl; len_out[1] = paired ? rl : 0;

    if (u_kind < 0.01) {
        /* unmapped first mate (random sequence) */
        for (uint32_t i = 0; i < rl; i++) { m1->seq[i] = "ACGT"[rng_below(&r, 4)]; if (paired) m2->seq[i] = "ACGT"[rng_below(&r, 4)]; }
        cb.n = 0;
        emit_rec(po, 0, 0, LCTY_FLAG_UNMAPPED, &cb);
        if (paired) emit_rec(po, 0, 0, LCTY_FLAG_UNMAPPED | LCTY_FLAG_MATE2, &cb);
        return;
    }
    if (u_kind < 0.02) {
        /* unrelated read: poor alignments only */
        for (uint32_t i = 0; i < rl; i++) { m1->seq[i] = "ACGT"[rng_below(&r, 4)]; if (paired) m2->seq[i] = "ACGT"[rng_below(&r, 4)]; }
        m1->len = rl; m2->len = rl;
        emit_variant_rec(L, &r, m1, rng_below(&r, A), fstart, rl / 4, 0, 0, S->ev, &cb, po);
        emit_variant_rec(L, &r, m1, rng_below(&r, A), fstart + 5, rl / 3, 0, LCTY_FLAG_SECONDARY, S->ev, &cb, po);
        if (paired) emit_variant_rec(L, &r, m2, rng_below(&r, A), fstart, rl / 4, 1, LCTY_FLAG_MATE2, S->ev, &cb, po);
        return;
    }

    if (paired) {
        uint32_t sa = fstart, sb = fstart + frag - rl;
        if (!flip) { make_mate(L, &r, h, sa, rl, 0, m1); make_mate(L, &r, h, sb, rl, 1, m2); }
        else       { make_mate(L, &r, h, sb, rl, 1, m1); make_mate(L, &r, h, sa, rl, 0, m2); }
    } else {
        make_mate(L, &r, h, fstart, rl, flip, m1);
    }
    int n_mates = paired ? 2 : 1;
    for (int e = 0; e < n_mates; e++) {
        mate_t* m = e ? m2 : m1;
        uint16_t ef = e ? LCTY_FLAG_MATE2 : 0;
        /* primary: the source haplotype */
        prepare_sites(L, m, h, S->ms);
        align_to(L, m, S->ms, h, h, ef, S->ev, &cb, po);
        double u = rng_unif(&r);
        if (u < 0.05) {         /* decoy >= 1 kb away, 2-9 mismatches */
            uint32_t a = rng_below(&r, A);
            uint32_t al = (uint32_t)(L->seq_off[a + 1] - L->seq_off[a]);
            uint32_t pos = (m->h_start + 1000 + rng_below(&r, al / 2)) % (al - rl - 2);
            emit_variant_rec(L, &r, m, a, pos, 2 + rng_below(&r, 8), rng_below(&r, 2), ef | LCTY_FLAG_SECONDARY, S->ev, &cb, po);
        } else if (u < 0.07) {  /* near-duplicate inside the same 128-bp bin on the source haplotype */
            emit_variant_rec(L, &r, m, h, b2a(L, h, a2b(L, h, m->h_start)) + 1 + rng_below(&r, 3), rng_below(&r, 3),
                             m->reverse, ef | LCTY_FLAG_SECONDARY, S->ev, &cb, po);
        }
        for (uint32_t a = 0; a < A; a++) {
            if (a == h) continue;
            align_to(L, m, S->ms, h, a, ef | LCTY_FLAG_SECONDARY, S->ev, &cb, po);
        }
    }
}

static scratch* scratch_new(uint32_t max_len) {
    scratch* S = (scratch*)malloc(sizeof(scratch));
    S->ev = (event*)malloc(sizeof(event) * MAX_EVENTS);
    S->cig = (uint32_t*)malloc(sizeof(uint32_t) * MAX_CIG);
    S->seq1 = (uint8_t*)malloc(max_len + 8); S->seq2 = (uint8_t*)malloc(max_len + 8);
    S->m1 = (mate_t*)malloc(sizeof(mate_t)); S->m2 = (mate_t*)malloc(sizeof(mate_t));
    S->ms = (mate_sites*)malloc(sizeof(mate_sites));
    return S;
}
static void scratch_free(scratch* S) { free(S->ev); free(S->cig); free(S->seq1); free(S->seq2); free(S->m1); free(S->m2); free(S->ms); free(S); }

static uint32_t max_read_len(const synth_locus* L) {
    return L->bg.is_paired ? L->read_len : (uint32_t)(2.5 * L->read_len) + 8;
}

/* Phase 1: per-pair sizes. mate_len[2n], rec_cnt[n], cig_cnt[n] */
void synth_reads_sizes(const synth_locus* L, uint64_t first, uint64_t n, uint32_t* mate_len,
                       uint32_t* rec_cnt, uint32_t* cig_cnt) {
#pragma omp parallel
    {
        scratch* S = scratch_new(max_read_len(L));
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < (int64_t)n; i++) {
            pair_out po; po.recs = NULL; po.cigar = NULL;
            gen_pair(L, first + (uint64_t)i, S, &po, mate_len + 2 * i);
            rec_cnt[i] = (uint32_t)po.n_recs; cig_cnt[i] = (uint32_t)po.n_cigar;
        }
        scratch_free(S);
    }
}

/* Phase 2: fill. All offset arrays (mate_off[2n+1], aln_off[n+1], cigar_off[n+1]) are inputs
 * computed by the caller from phase 1 (mate_off multiples of 32). bases2 / nmask must be zeroed. */
void synth_reads_fill(const synth_locus* L, uint64_t first, uint64_t n, const uint64_t* mate_off,
                      uint32_t* bases2, uint32_t* nmask, const uint64_t* aln_off, lcty_aln_rec* recs,
                      const uint64_t* cigar_off, uint32_t* cigar) {
#pragma omp parallel
    {
        scratch* S = scratch_new(max_read_len(L));
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < (int64_t)n; i++) {
            pair_out po; po.recs = recs + aln_off[i]; po.cigar = cigar + cigar_off[i];
            uint32_t lens[2];
            gen_pair(L, first + (uint64_t)i, S, &po, lens);
            for (int e = 0; e < 2; e++) {
                const uint8_t* s = e ? S->seq2 : S->seq1;
                uint64_t off = mate_off[2 * i + e];
                for (uint32_t j = 0; j < lens[e]; j++) {
                    uint64_t b = off + j;
                    uint32_t code; int isn = 0;
                    switch (s[j]) { case 'A': code = 0; break; case 'C': code = 1; break; case 'G': code = 2; break;
                                    case 'T': code = 3; break; default: code = 0; isn = 1; }
                    bases2[b >> 4] |= code << (2 * (b & 15));
                    if (isn) nmask[b >> 5] |= 1u << (b & 31);
                }
            }
        }
        scratch_free(S);
    }
}

/* ---- haplotype-to-haplotype alignments ---------------------------------------------------------------------------
 * The alignment of allele q (query) to allele r (reference) implied by the variants both were built from, as raw BAM
 * CIGAR words with =, X, I, D (what `haplotypes.paf` holds for a real database). Returns the number of words; words
 * beyond `cap` are counted but not written. n_matches / aln_len as in the PAF columns. */
typedef struct { uint32_t* w; uint32_t n, cap; uint32_t last_op, last_len; uint32_t n_matches, aln_len; } hapcig;
static void hc_flush(hapcig* h) {
    if (!h->last_len) return;
    if (h->n < h->cap) h->w[h->n] = (h->last_len << 4) | h->last_op;
    h->n++; h->last_len = 0;
}
static void hc_push(hapcig* h, uint32_t op, uint32_t len) {
    if (!len) return;
    if (op == 7) h->n_matches += len;
    h->aln_len += len;
    if (h->last_len && h->last_op == op) { h->last_len += len; return; }
    hc_flush(h);
    h->last_op = op; h->last_len = len;
}
uint32_t synth_hap_cigar(const synth_locus* L, uint32_t q, uint32_t r, uint32_t* words, uint32_t cap, uint32_t* n_matches, uint32_t* aln_len) {
    hapcig h; memset(&h, 0, sizeof(h)); h.w = words; h.cap = cap;
    uint32_t p = 0;
    for (uint32_t i = 0; i < L->n_vars; i++) {
        const variant* v = &L->vars[i];
        const int cq = carries(L, q, v), cr = carries(L, r, v);
        if (!cq && !cr) continue;
        if (v->pos < p) continue;
        hc_push(&h, 7, v->pos - p); p = v->pos;
        if (v->type == 0) { hc_push(&h, cq == cr ? 7 : 8, 1); p++; }
        else if (v->type == 1) {
            hc_push(&h, 7, 1); p++;
            hc_push(&h, cq && cr ? 7 : (cq ? 1 : 2), v->len);
        } else {
            if (cq && !cr) hc_push(&h, 2, v->len);          /* the query lacks the stretch */
            else if (cr && !cq) hc_push(&h, 1, v->len);
            p += v->len;
        }
    }
    hc_push(&h, 7, L->base_len - p);
    hc_flush(&h);
    if (n_matches) *n_matches = h.n_matches;
    if (aln_len) *aln_len = h.aln_len;
    return h.n;
}

/* The records of a chunk as lcty_aln_counted entries (harness side of lcty_reads_append_counted): what a caller does while it walks
 * raw_cigar() — count_region_operations_fast + limited_clipping (seq/aln.rs:288-317), hard_to_soft (seq/cigar.rs:309-320).
 * out[4 * n_recs] = {pos | flags << 28, contig | matches << 16, mismatches | insertions << 16, deletions | clipping << 16}.
 * Returns 0, or 1 + the index of the first record that cannot be counted (unsupported operation, count above 65535). */
uint64_t synth_count_records(uint64_t n_pairs, const uint64_t* aln_off, const lcty_aln_rec* recs, const uint64_t* cigar_off,
                             const uint32_t* cigar, const uint32_t* allele_len, uint32_t n_alleles, uint32_t* out) {
    uint64_t bad = 0;
#pragma omp parallel for schedule(static) reduction(max : bad)
    for (uint64_t r = 0; r < n_pairs; r++) {
        for (uint64_t i = aln_off[r]; i < aln_off[r + 1]; i++) {
            const lcty_aln_rec* rc = &recs[i];
            const uint32_t* cg = cigar + cigar_off[r] + rc->cigar_rel;
            uint32_t n = rc->n_cigar, matches = 0, mism = 0, ins = 0, del = 0, left = 0, right = 0;
            int ok = rc->contig < n_alleles || (rc->flags & LCTY_FLAG_UNMAPPED);
            for (uint32_t t = 0; t < n; t++) {
                uint32_t op = cg[t] & 15u, len = cg[t] >> 4;
                if (op == LCTY_CIGAR_H && (t == 0 || t + 1 == n)) op = LCTY_CIGAR_S;
                switch (op) {
                    case LCTY_CIGAR_EQ: matches += len; break;
                    case LCTY_CIGAR_X: mism += len; break;
                    case LCTY_CIGAR_I: ins += len; break;
                    case LCTY_CIGAR_D: del += len; break;
                    case LCTY_CIGAR_S: if (t == 0) left = len; else if (t + 1 == n) right = len; break;
                    default: ok = 0;
                }
            }
            const uint32_t clen = rc->contig < n_alleles ? allele_len[rc->contig] : 0;
            const uint32_t end = rc->pos + matches + mism + del;
            const uint32_t lc = left < rc->pos ? left : rc->pos, room = clen > end ? clen - end : 0;
            const uint32_t clip = lc + (right < room ? right : room);
            if (matches > 65535 || mism > 65535 || ins > 65535 || del > 65535 || clip > 65535 || rc->pos >= (1u << 28)) ok = 0;
            if (n == 0 && !(rc->flags & LCTY_FLAG_UNMAPPED)) ok = 0;
            if (!ok && i + 1 > bad) bad = i + 1;
            const uint32_t fl = ((rc->flags & LCTY_FLAG_REVERSE) ? 1u : 0u) | ((rc->flags & (LCTY_FLAG_SECONDARY | LCTY_FLAG_SUPPL)) ? 2u : 0u)
                              | ((rc->flags & LCTY_FLAG_UNMAPPED) ? 4u : 0u);
            out[4 * i] = (rc->pos & 0x0FFFFFFFu) | (fl << 28);
            out[4 * i + 1] = rc->contig | (matches << 16);
            out[4 * i + 2] = mism | (ins << 16);
            out[4 * i + 3] = del | (clip << 16);
        }
    }
    return bad;
}
