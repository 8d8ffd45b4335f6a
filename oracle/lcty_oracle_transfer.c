/* lcty_oracle_transfer.c — TEST INFRASTRUCTURE ONLY (see lcty_oracle.h): alignment recovery of AllAlignments::load,
 * SURVEY.md §8a rows a13-a14. PARITY UNPINNED, twice over: the reference has no tests, and the aligner it calls is
 * WFA2-lib (un-vendored, unpinned HEAD; call sites src/seq/wfa.rs:199, 206, 230-233, 346-351), whose published
 * algorithm is an exact gap-affine alignment — restated here as a Gotoh dynamic programme with a fixed tie rule
 * (walking back from the end: diagonal before deletion before insertion, a gap is extended before it is opened). WFA2's heuristics (adaptive band of 10
 * diagonals, wfa.rs:186-191) only matter for stretches whose optimal path leaves that band; its step limit
 * (accuracy 6 = 10 000, wfa.rs:103-117, 175) is restated as "optimal penalty above 10 000 -> align_simple".
 *
 * Restates: seq/cigar.rs:203-208, 323-375, 514-561 (Cigar), 969-986, 1099-1162 (CigarIndex), 1167-1237 (optimize),
 * 1248-1381 (transfer_alignment::<false> / transfer_read_alignment), 1422-1466 (double_cigar_move_and_shift);
 * seq/wfa.rs:30-100 (Penalties), 254-365 (align / smart_align / align_ends); seq/transfer.rs:21-140 (HapAlns). */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "lcty_oracle_internal.h"

#define MIN(a, b) ((a) < (b) ? (a) : (b))
#define MAX(a, b) ((a) > (b) ? (a) : (b))

/* ------------------------------------------------------------------ Cigar (BAM operation codes) */
enum { OP_M = 0, OP_I = 1, OP_D = 2, OP_S = 4, OP_H = 5, OP_EQ = 7, OP_X = 8 };

static int cons_q(uint32_t op) { return op == OP_M || op == OP_EQ || op == OP_X || op == OP_I || op == OP_S; }
static int cons_r(uint32_t op) { return op == OP_M || op == OP_EQ || op == OP_X || op == OP_D; }
static uint32_t op_invert(uint32_t op) {                 /* Operation::invert — cigar.rs:147-159 */
    if (op == OP_I || op == OP_S) return OP_D;
    if (op == OP_D) return OP_I;
    return op;
}

void orc_cigar_init(orc_cigar* c) { memset(c, 0, sizeof(*c)); }
void orc_cigar_free(orc_cigar* c) { free(c->t); memset(c, 0, sizeof(*c)); }
void orc_cigar_clear(orc_cigar* c) { c->n = 0; c->rlen = c->qlen = 0; }
static void cg_reserve(orc_cigar* c, uint32_t extra) {
    if (c->n + extra > c->cap) {
        c->cap = MAX(2 * c->cap, c->n + extra + 8);
        c->t = (orc_citem*)realloc(c->t, sizeof(orc_citem) * c->cap);
    }
}
/* push_unchecked — cigar.rs:343-352 */
void orc_cigar_push_unchecked(orc_cigar* c, uint32_t op, uint32_t len) {
    cg_reserve(c, 1);
    if (cons_q(op)) c->qlen += len;
    if (cons_r(op)) c->rlen += len;
    c->t[c->n].op = op; c->t[c->n].len = len; c->n++;
}
/* push_checked — cigar.rs:355-363 */
void orc_cigar_push_checked(orc_cigar* c, uint32_t op, uint32_t len) {
    if (cons_q(op)) c->qlen += len;
    if (cons_r(op)) c->rlen += len;
    if (c->n && c->t[c->n - 1].op == op) c->t[c->n - 1].len += len;
    else { cg_reserve(c, 1); c->t[c->n].op = op; c->t[c->n].len = len; c->n++; }
}
static void cg_append_items(orc_cigar* c, const orc_citem* it, uint32_t n) {      /* tuples.extend_from_slice: lengths untouched */
    if (!n) return;
    cg_reserve(c, n);
    memcpy(c->t + c->n, it, sizeof(orc_citem) * n);
    c->n += n;
}
void orc_cigar_copy(orc_cigar* dst, const orc_cigar* src) {
    orc_cigar_clear(dst);
    cg_reserve(dst, src->n);
    memcpy(dst->t, src->t, sizeof(orc_citem) * src->n);
    dst->n = src->n; dst->rlen = src->rlen; dst->qlen = src->qlen;
}
/* Cigar::from_raw — cigar.rs:296-302 (+ hard_to_soft 309-320 for the ends) */
void orc_cigar_from_raw(orc_cigar* c, const uint32_t* raw, uint32_t n, int hard_to_soft) {
    orc_cigar_clear(c);
    for (uint32_t i = 0; i < n; i++) {
        uint32_t op = raw[i] & 15u;
        if (hard_to_soft && op == OP_H && (i == 0 || i + 1 == n)) op = OP_S;
        orc_cigar_push_unchecked(c, op, raw[i] >> 4);
    }
}
uint32_t orc_cigar_to_raw(const orc_cigar* c, uint32_t* out) {
    for (uint32_t i = 0; i < c->n; i++) out[i] = (c->t[i].len << 4) | c->t[i].op;
    return c->n;
}
/* Cigar::invert — cigar.rs:323-329 */
static void cg_invert(orc_cigar* dst, const orc_cigar* src) {
    orc_cigar_copy(dst, src);
    for (uint32_t i = 0; i < dst->n; i++) dst->t[i].op = op_invert(dst->t[i].op);
    dst->rlen = src->qlen; dst->qlen = src->rlen;
}

/* ------------------------------------------------------------------ CigarIndex — cigar.rs:1085-1162 */
#define SPARSE_STEP_PWR 8u
#define SPARSE_MASK ((1u << SPARSE_STEP_PWR) - 1u)
typedef struct { uint32_t cigar_ix, pos; } sparse_ent;
typedef struct {
    uint32_t (*positions)[2];          /* [qpos, rpos] at the start of every item */
    sparse_ent* sparse[2]; uint32_t n_sparse[2];
} cigar_index;

static void sparse_update(sparse_ent** v, uint32_t* n, uint32_t* cap, uint32_t cigar_ix, uint32_t len, uint32_t pos1, uint32_t pos2,
                          int consumes_other) {            /* update_sparse_index — cigar.rs:972-986 */
    const uint32_t last = (pos1 + len - 1) >> SPARSE_STEP_PWR;
    for (uint32_t i = *n; i <= last; i++) {
        if (*n == *cap) { *cap = *cap ? 2 * *cap : 16; *v = (sparse_ent*)realloc(*v, sizeof(sparse_ent) * *cap); }
        const uint32_t sp1 = i << SPARSE_STEP_PWR;
        (*v)[*n].cigar_ix = cigar_ix;
        (*v)[*n].pos = pos2 + (consumes_other ? sp1 - pos1 : 0u);
        (*n)++;
    }
}
static void index_new(cigar_index* ix, const orc_cigar* c) {
    memset(ix, 0, sizeof(*ix));
    ix->positions = (uint32_t(*)[2])malloc(sizeof(uint32_t[2]) * MAX(c->n, 1));
    uint32_t qpos = 0, rpos = 0, cap[2] = {0, 0};
    for (uint32_t i = 0; i < c->n; i++) {
        const uint32_t op = c->t[i].op, len = c->t[i].len;
        ix->positions[i][0] = qpos; ix->positions[i][1] = rpos;
        const uint32_t old_q = qpos;
        if (cons_q(op)) { sparse_update(&ix->sparse[0], &ix->n_sparse[0], &cap[0], i, len, qpos, rpos, cons_r(op)); qpos += len; }
        if (cons_r(op)) { sparse_update(&ix->sparse[1], &ix->n_sparse[1], &cap[1], i, len, rpos, old_q, cons_q(op)); rpos += len; }
    }
    for (int d = 0; d < 2; d++) {                          /* final entry: last item, length of the other sequence */
        ix->sparse[d] = (sparse_ent*)realloc(ix->sparse[d], sizeof(sparse_ent) * (ix->n_sparse[d] + 1));
        ix->sparse[d][ix->n_sparse[d]].cigar_ix = c->n - 1;
        ix->sparse[d][ix->n_sparse[d]].pos = d == 0 ? c->rlen : c->qlen;
        ix->n_sparse[d]++;
    }
}
static void index_free(cigar_index* ix) { free(ix->positions); free(ix->sparse[0]); free(ix->sparse[1]); }

typedef struct { uint32_t min_cigar_ix, max_cigar_ix, approx_pos; } approx_position;
typedef struct { uint32_t cigar_ix, qpos_at_ix, rpos_at_ix; } cigar_offset;

/* find_approx_position — cigar.rs:1128-1140; dir 0 = QueryToRef, 1 = RefToQuery */
static approx_position find_approx_position(const cigar_index* ix, uint32_t qpos, int dir) {
    const sparse_ent* s = ix->sparse[dir];
    const uint32_t i = qpos >> SPARSE_STEP_PWR;
    approx_position a;
    a.min_cigar_ix = s[i].cigar_ix; a.max_cigar_ix = s[i + 1].cigar_ix;
    a.approx_pos = s[i].pos + (((qpos & SPARSE_MASK) * (s[i + 1].pos - s[i].pos)) >> SPARSE_STEP_PWR);
    return a;
}
/* find_cigar_offset — cigar.rs:1143-1162; bisect::right_by_at(positions, query_pos <=> qpos, lo, hi) - 1 */
static cigar_offset find_cigar_offset(const cigar_index* ix, uint32_t qpos, approx_position ap, int dir) {
    uint32_t ci;
    if (ap.min_cigar_ix == ap.max_cigar_ix) ci = ap.min_cigar_ix;
    else {
        uint32_t lo = ap.min_cigar_ix, hi = ap.max_cigar_ix + 1;        /* first index whose query position exceeds qpos */
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (ix->positions[mid][dir] <= qpos) lo = mid + 1; else hi = mid;
        }
        ci = lo - 1;
    }
    cigar_offset o;
    o.cigar_ix = ci; o.qpos_at_ix = ix->positions[ci][dir]; o.rpos_at_ix = ix->positions[ci][1 - dir];
    return o;
}

/* ------------------------------------------------------------------ the aligner — wfa.rs */
#define PEN_X 4
#define PEN_O 6
#define PEN_E 1
#define MAX_STEPS 10000               /* alignment_steps(6) — wfa.rs:103-117 */
#define DP_CELL_LIMIT (64u << 20)     /* beyond this the dynamic programme is not attempted (treated like a dropped alignment) */

/* Penalties::align_simple — wfa.rs:49-84 */
static int align_simple(const uint8_t* s1, uint32_t n, const uint8_t* s2, uint32_t m, orc_cigar* cg) {
    const int diff = (int)n - (int)m;
    int score;
    uint32_t i = 0, j = 0;
    if (diff < 0) { orc_cigar_push_unchecked(cg, OP_I, (uint32_t)(-diff)); score = -PEN_O + diff * PEN_E; j = (uint32_t)(-diff); }
    else if (diff > 0) { orc_cigar_push_unchecked(cg, OP_D, (uint32_t)diff); score = -PEN_O - diff * PEN_E; i = (uint32_t)diff; }
    else score = 0;
    int curr_match = s1[i] == s2[j];
    uint32_t curr_len = 1;
    for (uint32_t t = 1; i + t < n && j + t < m; t++) {
        const int eq = s1[i + t] == s2[j + t];
        if (eq != curr_match) {
            orc_cigar_push_unchecked(cg, curr_match ? OP_EQ : OP_X, curr_len);
            score -= curr_match ? 0 : PEN_X * (int)curr_len;
            curr_match = !curr_match; curr_len = 1;
        } else curr_len++;
    }
    orc_cigar_push_unchecked(cg, curr_match ? OP_EQ : OP_X, curr_len);
    score -= curr_match ? 0 : PEN_X * (int)curr_len;
    return score;
}

/* Gap-affine alignment of s1 (reference, length n) and s2 (query, length m) with a match bonus `mb` (0: global aligner,
 * 2: the semi-global aligner, wfa.rs:194-197). mode 0: end to end; 1: free begin of both (LEFT, 342-344); 2: free end of
 * both (RIGHT, 345-347). The operations are appended one base at a time with push_checked (wfa.rs:271-283); returns the
 * penalty, or DP_DROPPED when the alignment is "dropped" (status != 0, 262-266). */
typedef struct { int32_t m, d, i; } dpcell;      /* best penalty ending in a diagonal step / deletion (ref only) / insertion */
#define INF32 (1 << 28)
#define DP_DROPPED (-(1 << 30))          /* penalties can be negative with a match bonus */
static int dp_align(const uint8_t* s1, uint32_t n, const uint8_t* s2, uint32_t m, int mb, int mode, uint8_t* ops_out, uint32_t* n_ops) {
    if ((uint64_t)(n + 1) * (uint64_t)(m + 1) > DP_CELL_LIMIT) return DP_DROPPED;
    const uint32_t W = m + 1;
    dpcell* D = (dpcell*)malloc(sizeof(dpcell) * (size_t)(n + 1) * W);
#define AT(a, b) D[(size_t)(a) * W + (b)]
    for (uint32_t a = 0; a <= n; a++) {
        for (uint32_t b = 0; b <= m; b++) {
            dpcell c; c.m = c.d = c.i = INF32;
            if (a == 0 && b == 0) c.m = 0;
            else if (mode == 1 && (a == 0 || b == 0)) c.m = 0;            /* a prefix of one sequence is skipped for free */
            if (a > 0 && b > 0) {
                const dpcell p = AT(a - 1, b - 1);
                const int32_t best = MIN(p.m, MIN(p.d, p.i));
                if (best < INF32) { const int32_t v = best + (s1[a - 1] == s2[b - 1] ? -mb : PEN_X); if (v < c.m) c.m = v; }
            }
            if (a > 0) {
                const dpcell p = AT(a - 1, b);
                int32_t v = MIN(p.m, p.i) + PEN_O + PEN_E;
                if (p.d + PEN_E < v) v = p.d + PEN_E;
                if (v < INF32) c.d = v;
            }
            if (b > 0) {
                const dpcell p = AT(a, b - 1);
                int32_t v = MIN(p.m, p.d) + PEN_O + PEN_E;
                if (p.i + PEN_E < v) v = p.i + PEN_E;
                if (v < INF32) c.i = v;
            }
            AT(a, b) = c;
        }
    }
    uint32_t ea = n, eb = m;
    int32_t best = INF32;
    if (mode == 2) {                                                     /* the alignment may stop on the last row or column */
        for (uint32_t b = 0; b <= m; b++) { const dpcell c = AT(n, b); const int32_t v = MIN(c.m, MIN(c.d, c.i)); if (v < best) { best = v; ea = n; eb = b; } }
        for (uint32_t a = 0; a <= n; a++) { const dpcell c = AT(a, m); const int32_t v = MIN(c.m, MIN(c.d, c.i)); if (v < best) { best = v; ea = a; eb = m; } }
    } else { const dpcell c = AT(n, m); best = MIN(c.m, MIN(c.d, c.i)); }
    if (best >= INF32 || best > MAX_STEPS) { free(D); return DP_DROPPED; }
    /* walk back; state preference diagonal > deletion > insertion */
    uint32_t k = 0;
    uint8_t* rev = (uint8_t*)malloc((size_t)n + m + 2);
    if (mode == 2) {                                                     /* the skipped suffix, as WFA2 reports it */
        for (uint32_t b = m; b > eb; b--) rev[k++] = 'I';
        for (uint32_t a = n; a > ea; a--) rev[k++] = 'D';
    }
    uint32_t a = ea, b = eb;
    const dpcell ce = AT(a, b);
    int st = (ce.m <= ce.d && ce.m <= ce.i) ? 0 : (ce.d <= ce.i ? 1 : 2);
    while (a > 0 || b > 0) {
        if (mode == 1 && (a == 0 || b == 0) && st == 0) break;            /* reached the free border */
        const dpcell c = AT(a, b);
        if (st == 0) {
            const dpcell p = AT(a - 1, b - 1);
            const int32_t step = s1[a - 1] == s2[b - 1] ? -mb : PEN_X;
            rev[k++] = s1[a - 1] == s2[b - 1] ? '=' : 'X';
            const int32_t need = c.m - step;
            st = p.m == need ? 0 : (p.d == need ? 1 : 2);
            a--; b--;
        } else if (st == 1) {                                            /* a deletion ends here: extended, or opened */
            const dpcell p = AT(a - 1, b);
            rev[k++] = 'D';
            if (p.d + PEN_E == c.d) st = 1;
            else st = p.m <= p.i ? 0 : 2;
            a--;
        } else {
            const dpcell p = AT(a, b - 1);
            rev[k++] = 'I';
            if (p.i + PEN_E == c.i) st = 2;
            else st = p.m <= p.d ? 0 : 1;
            b--;
        }
    }
    if (mode == 1) {                                                     /* the skipped prefix, as WFA2 reports it */
        for (; b > 0; b--) rev[k++] = 'I';
        for (; a > 0; a--) rev[k++] = 'D';
    }
    for (uint32_t t = 0; t < k; t++) ops_out[t] = rev[k - 1 - t];
    *n_ops = k;
    free(rev); free(D);
#undef AT
    return best;
}

static uint32_t op_from_char(uint8_t ch) { return ch == '=' ? OP_EQ : ch == 'X' ? OP_X : ch == 'I' ? OP_I : OP_D; }

/* Aligner::align::<LEFT_CLIPPING> — wfa.rs:254-299. `semiglobal`: 0 global aligner, 1 LEFT, 2 RIGHT free ends */
static int aligner_align(const uint8_t* s1, uint32_t n, const uint8_t* s2, uint32_t m, int semiglobal, int left_clipping, orc_cigar* cg) {
    uint8_t* ops = (uint8_t*)malloc((size_t)n + m + 2);
    uint32_t n_ops = 0;
    const int pen = dp_align(s1, n, s2, m, semiglobal ? MAX(1, PEN_X / 2) : 0, semiglobal, ops, &n_ops);
    if (pen == DP_DROPPED) { free(ops); return align_simple(s1, n, s2, m, cg); }   /* status != 0 */
    int no_matches_yet = 1;
    for (uint32_t t = 0; t < n_ops; t++) {
        const uint32_t op = op_from_char(ops[t]);
        if (left_clipping && no_matches_yet && op == OP_EQ) {
            no_matches_yet = 0;
            const uint32_t soft = cg->qlen;
            orc_cigar_clear(cg);
            if (soft > 0) orc_cigar_push_unchecked(cg, OP_I, soft);
        }
        orc_cigar_push_checked(cg, op, 1);
    }
    if (left_clipping && no_matches_yet) {
        const uint32_t soft = cg->qlen;
        orc_cigar_clear(cg);
        if (soft > 0) orc_cigar_push_unchecked(cg, OP_I, soft);
    }
    free(ops);
    return -pen;
}

/* smart_align — wfa.rs:301-347; max_gap: 0xFFFFFFFF stands for the `()` threshold (never under) */
static int smart_align(const uint8_t* seq1, uint32_t i1, uint32_t i2, const uint8_t* seq2, uint32_t j1, uint32_t j2, uint32_t max_gap,
                       orc_cigar* cg) {
    const uint32_t jump1 = i2 - i1, jump2 = j2 - j1;
    if (jump1 > 0 && jump2 > 0) {
        const uint8_t* a = seq1 + i1; const uint8_t* b = seq2 + j1;
        const uint32_t safe_mismatch = (2 * PEN_O + 2 * PEN_E) / PEN_X;      /* wfa.rs:212 */
        if (max_gap < jump1 || max_gap < jump2) return align_simple(a, jump1, b, jump2, cg);
        if (jump1 == jump2 && jump1 <= safe_mismatch) {
            int ndiff = 0;
            for (uint32_t t = 0; t < jump1; t++) {
                orc_cigar_push_checked(cg, a[t] == b[t] ? OP_EQ : OP_X, 1);
                ndiff -= a[t] != b[t];
            }
            return ndiff * PEN_X;
        }
        return aligner_align(a, jump1, b, jump2, 0, 0, cg);
    }
    if (jump1 > 0) { orc_cigar_push_unchecked(cg, OP_D, jump1); return -PEN_O - (int)jump1 * PEN_E; }
    if (jump2 > 0) { orc_cigar_push_unchecked(cg, OP_I, jump2); return -PEN_O - (int)jump2 * PEN_E; }
    return 0;
}

/* align_ends::<LEFT> — wfa.rs:349-365 */
static void align_ends(int left, const uint8_t* seq1, uint32_t i1, uint32_t i2, const uint8_t* seq2, uint32_t j1, uint32_t j2, orc_cigar* cg) {
    if (i1 == i2) { orc_cigar_push_unchecked(cg, OP_I, j2 - j1); return; }
    aligner_align(seq1 + i1, i2 - i1, seq2 + j1, j2 - j1, left ? 1 : 2, left, cg);
    if (!left) {
        uint32_t soft = 0;
        while (cg->n && cg->t[cg->n - 1].op != OP_EQ) {                      /* pop_if(op != Equal) */
            const orc_citem it = cg->t[--cg->n];
            if (cons_q(it.op)) { cg->qlen -= it.len; soft += it.len; }
            if (cons_r(it.op)) cg->rlen -= it.len;
        }
        if (soft > 0) orc_cigar_push_unchecked(cg, OP_I, soft);
    }
}

static int g_optimize = 1;
void orc_transfer_set_optimize(int on) { g_optimize = on; }

/* ------------------------------------------------------------------ Cigar::optimize — cigar.rs:1167-1237 */
static void cigar_optimize(orc_cigar* self, const uint8_t* ref_seq, const uint8_t* query_seq, uint32_t max_gap, uint32_t anchor_size) {
    uint32_t i = 0, qpos1 = 0, rpos1 = 0, qpos2 = 0, rpos2 = 0;
    uint8_t flag = 0;
    orc_cigar nc; orc_cigar_init(&nc);
    int have = 0;
    for (uint32_t j = 0; j < self->n; j++) {
        const uint32_t op = self->t[j].op, len = self->t[j].len;
        const int cq = cons_q(op), cr = cons_r(op);
        if (cq && cr && len >= anchor_size) {
            const uint32_t qshift = qpos2 - qpos1, rshift = rpos2 - rpos1;
            if (flag == 3 && !(max_gap < qshift) && !(max_gap < rshift)) {
                if (!have) { have = 1; cg_append_items(&nc, self->t, i); nc.qlen = qpos1; nc.rlen = rpos1; }
                smart_align(ref_seq, rpos1, rpos2, query_seq, qpos1, qpos2, 0xFFFFFFFFu, &nc);
                i = j;
            }
            qpos2 += len; rpos2 += len; qpos1 = qpos2; rpos1 = rpos2; flag = 0;
            if (have) {
                cg_append_items(&nc, self->t + i, j - i);
                orc_cigar_push_checked(&nc, op, len);
                nc.qlen = qpos2; nc.rlen = rpos2;
            }
            i = j + 1;
        } else {
            qpos2 += cq ? len : 0; rpos2 += cr ? len : 0;
            flag |= (uint8_t)((cq ? 0 : 1) | ((cr ? 0 : 1) << 1));
        }
    }
    const uint32_t qshift = qpos2 - qpos1, rshift = rpos2 - rpos1;
    if (flag == 3 && !(max_gap < qshift) && !(max_gap < rshift)) {
        if (!have) { have = 1; cg_append_items(&nc, self->t, i); nc.qlen = qpos1; nc.rlen = rpos1; }
        smart_align(ref_seq, rpos1, rpos2, query_seq, qpos1, qpos2, 0xFFFFFFFFu, &nc);
        i = self->n;
    }
    if (have) {
        cg_append_items(&nc, self->t + i, self->n - i);
        orc_citem* old = self->t; const uint32_t oldcap = self->cap;
        self->t = nc.t; self->n = nc.n; self->cap = nc.cap;                  /* self.tuples = new_cigar.tuples (lengths stay) */
        nc.t = old; nc.cap = oldcap;
    }
    orc_cigar_free(&nc);
}

/* double_cigar_move_and_shift — cigar.rs:1422-1466. class: 0 both, 1 query only, 2 ref only */
static int cons_class(uint32_t op) { return cons_q(op) && cons_r(op) ? 0 : (cons_q(op) ? 1 : 2); }
static uint32_t double_move(uint32_t op1, uint32_t op2, uint32_t* pos1, uint32_t* rem1, uint32_t* pos2, uint32_t* rem2) {
    static const uint8_t T[3][3][4] = {      /* [op1 class][op2 class] = read_moves, read_cigar_shifts, hap_moves, hap_cigar_shifts */
        /* op1 both  */ {{1, 1, 1, 1}, {1, 1, 0, 1}, {0, 0, 1, 1}},
        /* op1 query */ {{1, 1, 0, 0}, {1, 1, 0, 0}, {1, 1, 1, 1}},
        /* op1 ref   */ {{0, 1, 1, 1}, {0, 1, 0, 1}, {0, 0, 1, 1}},
    };
    const uint8_t* f = T[cons_class(op1)][cons_class(op2)];
    const uint32_t shift = (f[1] && (!f[3] || *rem1 <= *rem2)) ? *rem1 : *rem2;
    *pos1 += f[0] ? shift : 0; *rem1 -= f[1] ? shift : 0;
    *pos2 += f[2] ? shift : 0; *rem2 -= f[3] ? shift : 0;
    return shift;
}

/* transfer_alignment::<false> — cigar.rs:1248-1368, as called by transfer_read_alignment (1371-1384): cigar_ij = read on hapQ
 * (direction QueryToRef), cigar_jk = hapQ vs hapT in direction `dir_jk` (0 QueryToRef, 1 RefToQuery), no maximum gap,
 * anchor size 5. Returns start_k; the new CIGAR in `out`. */
uint32_t orc_transfer_read_alignment(const orc_cigar* cigar_jk, int dir_jk, uint32_t start_j, uint32_t off_cigar_ix, uint32_t off_qpos,
                                     uint32_t off_rpos, const orc_cigar* cigar_ij, const uint8_t* seq_i, uint32_t len_i,
                                     const uint8_t* seq_k, uint32_t len_k, orc_cigar* out) {
    const uint32_t anchor_size = 5, ANCHOR_MARGIN = 5, CLIP_PADDING = 3, FULL_MATCH_PADDING = 3;
    orc_cigar_clear(out);
    uint32_t jk = off_cigar_ix;
    uint32_t op2 = dir_jk ? op_invert(cigar_jk->t[jk].op) : cigar_jk->t[jk].op;
    const uint32_t init_shift = start_j - off_qpos;
    uint32_t len2 = cigar_jk->t[jk].len, rem2 = len2 - init_shift;
    jk++;
    uint32_t start_k = off_rpos + (cons_r(op2) ? init_shift : 0);
    if (op2 == OP_EQ && init_shift >= FULL_MATCH_PADDING && rem2 >= cigar_ij->rlen + FULL_MATCH_PADDING) {
        orc_cigar_copy(out, cigar_ij);
        return start_k;
    }
    uint32_t ij = 0;
    uint32_t len1 = cigar_ij->t[0].len, rem1 = len1, op1 = cigar_ij->t[0].op;
    ij++;
    uint32_t last1 = 0, pos1 = 0, last2 = start_k, pos2 = start_k;
    for (;;) {
        int add = -1;
        const int e1 = op1 == OP_EQ, e2 = op2 == OP_EQ;
        if (e1 && e2) { if (MIN(rem1, rem2) >= anchor_size) add = OP_EQ; }
        else if (e1 && !e2) { if (rem1 >= anchor_size && len1 - rem1 >= ANCHOR_MARGIN) add = (int)op2; }
        else if (!e1 && e2) { if (rem2 >= anchor_size && len2 - rem2 >= ANCHOR_MARGIN) add = (int)op1; }
        if (add >= 0) {
            if (last1 == 0 && pos1 > 0) {
                const uint32_t from = last2 > pos1 + CLIP_PADDING ? last2 - (pos1 + CLIP_PADDING) : 0;     /* saturating_sub */
                align_ends(1, seq_k, from, pos2, seq_i, last1, pos1, out);
                start_k = start_k + pos2 - last2 - out->rlen;
            } else smart_align(seq_k, last2, pos2, seq_i, last1, pos1, 0xFFFFFFFFu, out);
        }
        const uint32_t shift = double_move(op1, op2, &pos1, &rem1, &pos2, &rem2);
        if (add >= 0) { orc_cigar_push_checked(out, (uint32_t)add, shift); last1 = pos1; last2 = pos2; }
        if (rem1 == 0) {
            if (ij == cigar_ij->n) break;
            len1 = cigar_ij->t[ij].len; rem1 = len1; op1 = cigar_ij->t[ij].op; ij++;
        }
        if (rem2 == 0) {
            if (jk == cigar_jk->n) break;
            len2 = cigar_jk->t[jk].len; rem2 = len2; op2 = dir_jk ? op_invert(cigar_jk->t[jk].op) : cigar_jk->t[jk].op; jk++;
        }
    }
    if (last1 != len_i)
        align_ends(0, seq_k, last2, MIN(len_k, last2 + len_i - last1 + CLIP_PADDING), seq_i, last1, len_i, out);
    /* assert_eq!(len_i, new_cigar.qlen) — cigar.rs:1353 */
    /* MAX_OPTIMIZATION_GAP = 20, OPTIMIZATION_ANCHOR = 5. As upstream (cigar.rs:1362-1364, 1186-1196): optimize() indexes the
     * reference sequence with positions counted from the START OF THE CIGAR while it is handed the whole target contig, so a
     * stretch with both an insertion and a deletion between two anchors is realigned against target[rpos..] instead of
     * target[start_k + rpos..]. Kept as it is written; orc_transfer_set_optimize(0) switches the step off for tests of the
     * rest. */
    if (g_optimize) cigar_optimize(out, seq_k, seq_i, 20, 5);
    if (out->n) {                                                             /* boundary_ins_to_soft — cigar.rs:554-561 */
        if (out->t[0].op == OP_I) out->t[0].op = OP_S;
        if (out->t[out->n - 1].op == OP_I) out->t[out->n - 1].op = OP_S;
    }
    return start_k;
}

/* ------------------------------------------------------------------ HapAlns — transfer.rs:21-67 */
typedef struct { orc_cigar cigar; cigar_index index; int present; } hap_cell;
typedef struct { uint32_t id, n_matches; uint32_t order; } best_ix;
struct orc_hap_alns {
    uint32_t n_contigs, transfer_fails;
    double max_div;
    hap_cell* cells;                  /* [i * n + j], i < j */
    best_ix** best; uint32_t* n_best;
};

orc_hap_alns* orc_hap_alns_new(uint32_t n_contigs, uint32_t transfer_fails, double max_div) {
    orc_hap_alns* h = (orc_hap_alns*)calloc(1, sizeof(*h));
    h->n_contigs = n_contigs; h->transfer_fails = transfer_fails; h->max_div = max_div;
    h->cells = (hap_cell*)calloc((size_t)n_contigs * n_contigs, sizeof(hap_cell));
    h->best = (best_ix**)calloc(n_contigs, sizeof(best_ix*));
    h->n_best = (uint32_t*)calloc(n_contigs, sizeof(uint32_t));
    return h;
}
void orc_hap_alns_free(orc_hap_alns* h) {
    if (!h) return;
    for (size_t c = 0; c < (size_t)h->n_contigs * h->n_contigs; c++)
        if (h->cells[c].present) { orc_cigar_free(&h->cells[c].cigar); index_free(&h->cells[c].index); }
    for (uint32_t i = 0; i < h->n_contigs; i++) free(h->best[i]);
    free(h->cells); free(h->best); free(h->n_best); free(h);
}
/* HapAlns::add — transfer.rs:41-62 for an entry that is a full alignment on the forward strand; id1 = query, id2 = target */
void orc_hap_alns_add(orc_hap_alns* h, uint32_t id1, uint32_t id2, const uint32_t* raw, uint32_t n_raw, uint32_t n_matches, uint32_t aln_len) {
    const uint32_t a = MIN(id1, id2), b = MAX(id1, id2);
    hap_cell* cell = &h->cells[(size_t)a * h->n_contigs + b];
    if (cell->present || id1 == id2) return;
    const double div = aln_len == 0 ? INFINITY : (double)(aln_len - n_matches) / (double)aln_len;      /* paf.rs:201-208 */
    if (div > h->max_div) return;
    orc_cigar c; orc_cigar_init(&c);
    orc_cigar_from_raw(&c, raw, n_raw, 0);
    orc_cigar_init(&cell->cigar);
    if (id1 > id2) cg_invert(&cell->cigar, &c); else orc_cigar_copy(&cell->cigar, &c);
    orc_cigar_free(&c);
    index_new(&cell->index, &cell->cigar);
    cell->present = 1;
    for (int t = 0; t < 2; t++) {
        const uint32_t me = t ? id2 : id1, other = t ? id1 : id2;
        h->best[me] = (best_ix*)realloc(h->best[me], sizeof(best_ix) * (h->n_best[me] + 1));
        h->best[me][h->n_best[me]].id = other; h->best[me][h->n_best[me]].n_matches = n_matches;
        h->best[me][h->n_best[me]].order = h->n_best[me];
        h->n_best[me]++;
    }
}
static int cmp_best(const void* x, const void* y) {          /* sort_by(b.1.cmp(a.1)): stable, most matches first */
    const best_ix* a = (const best_ix*)x; const best_ix* b = (const best_ix*)y;
    if (a->n_matches != b->n_matches) return a->n_matches > b->n_matches ? -1 : 1;
    return a->order < b->order ? -1 : (a->order > b->order ? 1 : 0);
}
void orc_hap_alns_sort(orc_hap_alns* h) {
    for (uint32_t i = 0; i < h->n_contigs; i++) qsort(h->best[i], h->n_best[i], sizeof(best_ix), cmp_best);
}

uint32_t orc_hap_alns_n_best(const orc_hap_alns* h, uint32_t contig) { return h->n_best[contig]; }
uint32_t orc_hap_alns_best(const orc_hap_alns* h, uint32_t contig, uint32_t i) { return h->best[contig][i].id; }
uint32_t orc_hap_alns_transfer_fails(const orc_hap_alns* h) { return h->transfer_fails; }

/* one transfer of transfer_alignments' inner loop (transfer.rs:90-126) up to the new CIGAR: returns 1 and fills
 * (approx_pos) first; the caller probes its position collection with it and, if that misses, calls orc_hap_alns_transfer */
uint32_t orc_hap_alns_approx_pos(const orc_hap_alns* h, uint32_t source, uint32_t target, uint32_t source_start) {
    const hap_cell* cell = &h->cells[(size_t)MIN(source, target) * h->n_contigs + MAX(source, target)];
    return find_approx_position(&cell->index, source_start, source < target ? 0 : 1).approx_pos;
}
uint32_t orc_hap_alns_transfer(const orc_hap_alns* h, uint32_t source, uint32_t target, uint32_t source_start, const orc_cigar* source_cigar,
                               const uint8_t* read_seq, uint32_t read_len, const uint8_t* target_seq, uint32_t target_len, orc_cigar* out) {
    const hap_cell* cell = &h->cells[(size_t)MIN(source, target) * h->n_contigs + MAX(source, target)];
    const int dir = source < target ? 0 : 1;
    const approx_position ap = find_approx_position(&cell->index, source_start, dir);
    const cigar_offset off = find_cigar_offset(&cell->index, source_start, ap, dir);
    return orc_transfer_read_alignment(&cell->cigar, dir, source_start, off.cigar_ix, off.qpos_at_ix, off.rpos_at_ix, source_cigar, read_seq,
                                       read_len, target_seq, target_len, out);
}

/* test hook: the aligner on its own. mode 0 global, 1 free begin (LEFT), 2 free end (RIGHT); returns the penalty or -1 */
int orc_dp_align(const uint8_t* s1, uint32_t n, const uint8_t* s2, uint32_t m, int match_bonus, int mode, uint32_t* out_cigar, uint32_t out_cap,
                 uint32_t* n_out) {
    uint8_t* ops = (uint8_t*)malloc((size_t)n + m + 2);
    uint32_t n_ops = 0;
    const int pen = dp_align(s1, n, s2, m, match_bonus, mode, ops, &n_ops);
    orc_cigar c; orc_cigar_init(&c);
    if (pen != DP_DROPPED) for (uint32_t t = 0; t < n_ops; t++) orc_cigar_push_checked(&c, op_from_char(ops[t]), 1);
    *n_out = c.n;
    if (c.n <= out_cap) orc_cigar_to_raw(&c, out_cigar);
    orc_cigar_free(&c); free(ops);
    return pen;
}
