// lcty_transfer_device.hpp — device side of alignment recovery (K6): one lane carries one transfer of a read alignment
// from the haplotype it was found on to another haplotype, through the alignment of the two haplotypes
// (Cigar::transfer_read_alignment, src/seq/cigar.rs:1248-1384) with the aligner calls of src/seq/wfa.rs restated as an
// exact gap-affine dynamic programme (WFA2-lib computes the same optimum; its tie-breaks are not pinned anywhere, ours
// are: walking back from the end, diagonal before deletion before insertion, a gap is extended before it is opened).
// Scalar code per lane with private scratch in global memory: this is control-heavy work, the parallelism is across the
// (alignment, target haplotype) pairs of a read.
#pragma once

#include "lcty_device.hpp"

namespace lcty {
namespace xfer {

constexpr uint32_t OP_I = 1, OP_D = 2, OP_S = 4, OP_H = 5, OP_EQ = 7, OP_X = 8;
constexpr int PEN_X = 4, PEN_O = 6, PEN_E = 1;             // Penalties::default (wfa.rs:30-38)
constexpr int MAX_STEPS = 10000;                            // alignment_steps(6), wfa.rs:103-117
constexpr int INF32 = 1 << 28;
constexpr int DP_DROPPED = -(1 << 30);
constexpr int DP_TOO_BIG = DP_DROPPED + 1;                  // the stretch does not fit the lane's scratch at this level

// What a lane's scratch holds. The transfer kernel runs in levels: a read pair whose transfers need more than the level
// offers is handed to the next level (fewer lanes in flight, larger scratch); beyond the last level LCTY_ERR_UNSUPPORTED.
struct Limits {
    uint32_t cigar_cap;        // items of a transferred CIGAR
    uint32_t dp_dim;           // longest side of a stretch the aligner takes
    uint32_t dp_cells;         // (n + 1) * (m + 1) direction bytes
};
__host__ __device__ inline size_t lane_scratch_bytes(const Limits& m) {
    size_t b = static_cast<size_t>(m.cigar_cap) * 4 * 2;                       // two CIGARs, an item a word (length << 4 | operation)
    b += 3 * (static_cast<size_t>(m.dp_dim) + 1) * 12;                         // two rolling rows + the last column
    b += (2 * static_cast<size_t>(m.dp_dim) + 4 + 15) & ~size_t(15);           // operations of one alignment
    b += m.dp_cells;
    b += (static_cast<size_t>(m.cigar_cap) / 2 + 8) * 16;                      // the stretches a walk leaves for the aligner (Walk::n_jobs)
    return (b + 15) & ~size_t(15);
}

// consumes the query: M I S = X; consumes the reference: M D = X (bit `op` of a mask; operations are 4-bit codes)
__device__ __forceinline__ bool cons_q(uint32_t op) { return ((0x193u >> (op & 15u)) & 1u) != 0; }
__device__ __forceinline__ bool cons_r(uint32_t op) { return ((0x185u >> (op & 15u)) & 1u) != 0; }
__device__ __forceinline__ uint32_t op_invert(uint32_t op) {                  // Operation::invert, cigar.rs:147-159
    return (op == OP_I || op == OP_S) ? OP_D : (op == OP_D ? OP_I : op);
}

// Cigar under construction (cigar.rs:203-208): items {op, len}, an item one word, length << 4 | operation (the BAM form, which is also
// what leaves the kernel).
// The LAST item lives in registers (`pend`): a push that extends it — most pushes of a walk do — is an addition, and a push that starts a
// new item only hands the finished one on. Readers of the items (get / word / n) call flush() first.
// Layout in the lane's scratch: CHUNKS of 16 items, 64 bytes, chunk c of lane l at t[c * 1024 + l * 16] — and a chunk is written at once:
// the items of the chunk under construction are collected in LDS (`wb`, word k of lane l at wb[k * 64 + l]) and leave as four 16-byte
// stores when the 16th arrives. (Rounds 3-4 had item i of lane l at t[i * 64 + l], stored one by one: the lanes of a wavefront walk
// their transfers at their own pace, so every item was a 4-byte store into a 256-byte row of its own — 3 passes x 900 items x 64
// lanes of partial-line writes per chunk of targets, 28 KB written per transfer for 11, and every pass waited on them.)
// A lane has TWO such LDS slices of 16 words. Only one CIGAR of a lane is under construction at a time (the walk's, then the assembled one,
// then the optimized one) and at most one finished CIGAR is being read meanwhile: the new CIGAR collects its chunk in the slice the
// finished one does not own. finish(): a CIGAR of at most 16 items — every transferred short read — simply STAYS in its slice (`keep`) and
// never sees global memory until it is copied out; a longer one writes its last, partial chunk out, is read from global memory from then
// on, and its slice serves the reader as the window its items are read ahead into (ItemWindow).
constexpr uint32_t CIGAR_CHUNK = 16;
struct DCigar {
    uint32_t* t;           // global part: [cap / 16] chunks x 64 lanes x 16 words, already offset by the lane (lane * 16 words)
    uint32_t* wb;          // LDS: the chunk under construction, already offset by the lane (stride 64); nullptr once finished
    uint32_t* keep;        // LDS: ALL items of a finished CIGAR of at most 16 (then nothing of it is in global memory); else nullptr
    uint32_t* slice;       // the LDS slice this CIGAR was given (wb while under construction; `keep` or free for the reader's window afterwards)
    uint32_t n, rlen, qlen, cap;
    uint2 pend; bool has_pend;                                               // the item behind item n - 1, not stored yet
    bool overflow;                                                           // items were dropped: the lengths are still right
    __device__ void init(uint32_t* buf, uint32_t capacity, uint32_t* lds_slice) {
        t = buf; cap = capacity; wb = slice = lds_slice; keep = nullptr; n = 0; rlen = qlen = 0; overflow = false;
        pend = make_uint2(0, 0); has_pend = false;
    }
    __device__ __forceinline__ uint32_t* at(uint32_t i) const { return t + static_cast<size_t>(i >> 4) * (CIGAR_CHUNK * 64) + (i & 15u); }
    __device__ __forceinline__ bool in_wb(uint32_t i) const { return wb != nullptr && i >= (n & ~15u); }
    __device__ __forceinline__ uint32_t word(uint32_t i) const { return in_wb(i) ? wb[(i & 15u) * 64] : keep ? keep[(i & 15u) * 64] : *at(i); }
    __device__ __forceinline__ uint2 get(uint32_t i) const { const uint32_t w = word(i); return make_uint2(w & 15u, w >> 4); }
    __device__ __forceinline__ void set(uint32_t i, uint2 v) {
        const uint32_t w = (v.y << 4) | v.x;
        if (in_wb(i)) wb[(i & 15u) * 64] = w; else if (keep) keep[(i & 15u) * 64] = w; else *at(i) = w;
    }
    __device__ __forceinline__ void store_chunk(uint32_t first) {                // the 16 words of the LDS buffer -> chunk first / 16
        uint4* dst = reinterpret_cast<uint4*>(at(first));
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) dst[q] = make_uint4(wb[(4 * q) * 64], wb[(4 * q + 1) * 64], wb[(4 * q + 2) * 64], wb[(4 * q + 3) * 64]);
    }
    __device__ __forceinline__ void append(uint32_t w) {                         // n < cap
        wb[(n & 15u) * 64] = w; n++;
        if ((n & 15u) == 0) store_chunk(n - CIGAR_CHUNK);
    }
    // the last item taken off again (align_ends::<RIGHT>); stepping back over a chunk boundary brings that chunk back into the buffer
    __device__ __forceinline__ uint2 pop_back() {
        n--;
        if ((n & 15u) == 15u) {
            const uint4* src = reinterpret_cast<const uint4*>(at(n & ~15u));
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) {
                const uint4 v = src[q];
                wb[(4 * q) * 64] = v.x; wb[(4 * q + 1) * 64] = v.y; wb[(4 * q + 2) * 64] = v.z; wb[(4 * q + 3) * 64] = v.w;
            }
        }
        const uint32_t w = wb[(n & 15u) * 64];
        return make_uint2(w & 15u, w >> 4);
    }
    __device__ void clear() { n = 0; rlen = qlen = 0; has_pend = false; }
    __device__ __forceinline__ void flush() {
        if (has_pend) { if (n < cap) append((pend.y << 4) | pend.x); else overflow = true; has_pend = false; }
    }
    // nothing more will be pushed: a CIGAR of one chunk stays where it is, a longer one goes to global memory completely
    __device__ __forceinline__ void finish() {
        flush();
        if (wb) {
            if (n <= CIGAR_CHUNK) keep = wb;
            else if (n & 15u) store_chunk(n & ~15u);
            wb = nullptr;
        }
    }
    __device__ __forceinline__ void push_raw(uint2 it) { flush(); pend = it; has_pend = true; }          // lengths untouched
    __device__ __forceinline__ void push_unchecked(uint32_t op, uint32_t len) {   // cigar.rs:343-352
        if (cons_q(op)) qlen += len;
        if (cons_r(op)) rlen += len;
        push_raw(make_uint2(op, len));
    }
    __device__ __forceinline__ void push_checked(uint32_t op, uint32_t len) {     // cigar.rs:355-363
        if (cons_q(op)) qlen += len;
        if (cons_r(op)) rlen += len;
        if (has_pend && pend.x == op) { pend.y += len; return; }
        push_raw(make_uint2(op, len));
    }
};

// The next items of a lane's finished CIGAR, read ahead into LDS. A pass over a CIGAR in the lane's scratch (assemble, optimize) paid one
// trip to the L2 per item — every lane of the wavefront at its own item, nothing else to do meanwhile, 1.5-2.5 us each next to the
// stores of the 3 000 other wavefronts; a window is filled by ITEM_WIN loads that are in flight together. Filled by all lanes of a
// wavefront at the same point of the code (their positions differ, their pace does not): word k of lane l at w[k * 64 + l].
constexpr uint32_t ITEM_WIN = 16;
struct ItemWindow {
    uint32_t* w;           // LDS, already offset by the lane: the slice of the CIGAR that is read (free once that CIGAR is in global memory)
    uint32_t base, end;    // the window holds the items [base, end)
    __device__ __forceinline__ void init(const DCigar& c) { w = c.slice; base = end = 0; }
    __device__ __forceinline__ bool has(uint32_t i) const { return i - base < end - base; }
    __device__ __forceinline__ void fill(const DCigar& c, uint32_t from) {
        if (c.keep) { base = 0; end = c.n; return; }                          // the whole CIGAR is in this very slice: nothing to load
        base = from; end = min(from + ITEM_WIN, c.n);
        uint32_t v[ITEM_WIN];
#pragma unroll
        for (uint32_t k = 0; k < ITEM_WIN; k++) v[k] = from + k < c.n ? c.word(from + k) : 0u;
#pragma unroll
        for (uint32_t k = 0; k < ITEM_WIN; k++) w[k * 64] = v[k];
    }
    __device__ __forceinline__ uint2 get(uint32_t i) const { const uint32_t x = w[(i - base) * 64]; return make_uint2(x & 15u, x >> 4); }
    // item i of `c`: out of the window where it holds it
    __device__ __forceinline__ uint2 item(const DCigar& c, uint32_t i) const { return has(i) ? get(i) : c.get(i); }
};
// items [from, to) of `src` appended to `dst` as they are (push_raw), eight loads in flight at a time (one by one every load waited for
// the store in front of it)
__device__ inline void copy_items(const DCigar& src, uint32_t from, uint32_t to, DCigar& dst) {
    for (uint32_t k = from; k < to; k += 8) {
        uint32_t v[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) v[u] = k + u < to ? src.word(k + u) : 0u;
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) if (k + u < to) dst.push_raw(make_uint2(v[u] & 15u, v[u] >> 4));
    }
}

// the read in the orientation of the alignment (MateData::get_seq, locs.rs:87-93) and the target haplotype, as ASCII
struct Seqs {
    const uint64_t* w64; const uint32_t* nm; uint32_t read_len; bool flip;   // packed mate, N mask; flip = reverse complement
    const uint8_t* target; uint32_t target_len;
    __device__ __forceinline__ uint8_t q(uint32_t i) const {
        const uint32_t j = flip ? read_len - 1 - i : i;
        if ((nm[j >> 5] >> (j & 31u)) & 1u) return 'N';
        uint32_t c = static_cast<uint32_t>(w64[j >> 5] >> ((j & 31u) * 2u)) & 3u;
        if (flip) c = 3u - c;
        return static_cast<uint8_t>("ACGT"[c]);
    }
    __device__ __forceinline__ uint8_t r(uint32_t i) const { return target[i]; }
};

// per-lane scratch. The aligner's arrays are interleaved across the 64 lanes of the wavefront (element i of lane l at
// base[i * 64 + l]): the lanes of a wavefront align similar stretches in lockstep — the targets of one source alignment — and
// then touch the same element index at the same time, i.e. one contiguous line instead of 64 scattered ones
constexpr uint32_t LANE_STRIDE = 64;                                  // lane stride of the interleaved arrays
struct Scratch {
    uint32_t* cig_a;       // [cigar_cap / 16] chunks x LANE_STRIDE x 16 words (DCigar)
    uint32_t* cig_b;       // the same (assemble, optimize)
    uint8_t* ops;          // [2 * dp_dim + 4] x LANE_STRIDE            aligner output, reversed
    uint8_t* dirs;         // [dp_cells] x LANE_STRIDE
    uint4* jobs;           // [cigar_cap / 2 + 8] x LANE_STRIDE: {i1, n, j1, m} of the stretches between anchors a walk left for the aligner
    uint32_t* cig_free;    // the CIGAR buffer that is not in use (cig_b, or cig_a once the assembled CIGAR lives in cig_b)
    int32_t* rows;         // [2][dp_dim + 1][3] x LANE_STRIDE
    int32_t* lastcol;      // [dp_dim + 1][3] x LANE_STRIDE
    Limits lim;
    uint32_t big;          // bit 0: an end-to-end stretch was too big (placeholder of the right lengths), bit 1: an ends-free one
    unsigned long long cells;   // cells of the aligner's matrices this lane has filled (GCUPS of lcty_recover_alignments)
};

// `base`: the wavefront's block of 64 * lane_scratch_bytes(lim) bytes
__device__ inline Scratch scratch_at(uint8_t* base, uint32_t lane, const Limits& lim) {
    Scratch s;
    const size_t cig = static_cast<size_t>(lim.cigar_cap) * 4, row = (static_cast<size_t>(lim.dp_dim) + 1) * 12;
    const size_t nops = (2 * static_cast<size_t>(lim.dp_dim) + 4 + 15) & ~size_t(15);
    s.cig_a = reinterpret_cast<uint32_t*>(base) + lane * CIGAR_CHUNK; base += 64 * cig;        // cigar_cap is a multiple of the chunk
    s.cig_b = reinterpret_cast<uint32_t*>(base) + lane * CIGAR_CHUNK; base += 64 * cig;
    s.rows = reinterpret_cast<int32_t*>(base) + lane; base += 64 * 2 * row;
    s.lastcol = reinterpret_cast<int32_t*>(base) + lane; base += 64 * row;
    s.ops = base + lane; base += 64 * nops;
    s.dirs = base + lane; base += 64 * static_cast<size_t>(lim.dp_cells);
    s.jobs = reinterpret_cast<uint4*>(base) + lane;
    s.cig_free = s.cig_b;
    s.lim = lim; s.big = 0; s.cells = 0;
    return s;
}

// Penalties::align_simple (wfa.rs:49-84): reference [i1, i1+n) against query [j1, j1+m)
__device__ inline int align_simple(const Seqs& S, uint32_t i1, uint32_t n, uint32_t j1, uint32_t m, DCigar& cg) {
    const int diff = static_cast<int>(n) - static_cast<int>(m);
    int score;
    uint32_t i = 0, j = 0;
    if (diff < 0) { cg.push_unchecked(OP_I, static_cast<uint32_t>(-diff)); score = -PEN_O + diff * PEN_E; j = static_cast<uint32_t>(-diff); }
    else if (diff > 0) { cg.push_unchecked(OP_D, static_cast<uint32_t>(diff)); score = -PEN_O - diff * PEN_E; i = static_cast<uint32_t>(diff); }
    else score = 0;
    bool curr_match = S.r(i1 + i) == S.q(j1 + j);
    uint32_t curr_len = 1;
    for (uint32_t t = 1; i + t < n && j + t < m; t++) {
        const bool eq = S.r(i1 + i + t) == S.q(j1 + j + t);
        if (eq != curr_match) {
            cg.push_unchecked(curr_match ? OP_EQ : OP_X, curr_len);
            score -= curr_match ? 0 : PEN_X * static_cast<int>(curr_len);
            curr_match = !curr_match; curr_len = 1;
        } else curr_len++;
    }
    cg.push_unchecked(curr_match ? OP_EQ : OP_X, curr_len);
    score -= curr_match ? 0 : PEN_X * static_cast<int>(curr_len);
    return score;
}

// Gap-affine alignment of reference [i1, i1+n) and query [j1, j1+m); mb = match bonus (0 global aligner, 2 semi-global one,
// wfa.rs:194-197); mode 0 end to end, 1 free begin of both (LEFT), 2 free end of both (RIGHT). Writes the operations in
// REVERSE order into sc.ops and returns the penalty, or DP_DROPPED (wfa.rs:262-266: status != 0).
__device__ inline int dp_align(const Seqs& S, uint32_t i1, uint32_t n, uint32_t j1, uint32_t m, int mb, int mode, Scratch& sc, uint32_t* n_ops) {
    // end to end, every alignment pays for the difference of the lengths: beyond MAX_STEPS the aligner gives up whatever the bases are
    if (mode == 0 && n != m && PEN_O + static_cast<int>(n > m ? n - m : m - n) * PEN_E > MAX_STEPS) return DP_DROPPED;
    if (n > sc.lim.dp_dim || m > sc.lim.dp_dim || (static_cast<uint64_t>(n) + 1) * (m + 1) > sc.lim.dp_cells) {
        sc.big |= mode == 0 ? 1u : 2u;
        return DP_TOO_BIG;
    }
    const uint32_t W = m + 1;
    sc.cells += static_cast<unsigned long long>(n + 1) * W;
    int32_t* prev = sc.rows;
    int32_t* cur = sc.rows + static_cast<size_t>(sc.lim.dp_dim + 1) * 3 * LANE_STRIDE;
    // the bases of a short query stretch (the rule: a few bases between two anchors) are taken out of the packed read once, not once per cell
    uint64_t qpack = 0;
    const bool q_packed = m <= 8;
    if (q_packed) for (uint32_t b = 0; b < m; b++) qpack |= static_cast<uint64_t>(S.q(j1 + b)) << (8 * b);
    for (uint32_t a = 0; a <= n; a++) {
        const uint8_t rbase = a > 0 ? S.r(i1 + a - 1) : 0;
        // the cell to the left (this row) and the cell above-left (previous row) travel in registers: reading them back from the
        // rows in memory made every cell wait for the stores of the cell before it
        int32_t lm = INF32, ld = INF32, li = INF32;                            // cell (a, b - 1)
        int32_t gm = INF32, gd = INF32, gi = INF32;                            // cell (a - 1, b - 1)
        for (uint32_t b = 0; b <= m; b++) {
            int32_t cm = INF32, cd = INF32, ci = INF32;
            uint32_t dm = 3, dd = 0, di = 0;
            int32_t um = INF32, ud = INF32, ui = INF32;                        // cell (a - 1, b)
            if (a > 0) { um = prev[(b * 3) * LANE_STRIDE]; ud = prev[(b * 3 + 1) * LANE_STRIDE]; ui = prev[(b * 3 + 2) * LANE_STRIDE]; }
            if (a == 0 && b == 0) cm = 0;
            else if (mode == 1 && (a == 0 || b == 0)) cm = 0;                 // a prefix of one sequence is skipped for free
            if (a > 0 && b > 0) {
                const int32_t pm = gm, pd = gd, pi = gi;
                const int32_t best = min(pm, min(pd, pi));
                if (best < INF32) {
                    const uint8_t qbase = q_packed ? static_cast<uint8_t>(qpack >> (8 * (b - 1))) : S.q(j1 + b - 1);
                    const int32_t v = best + (rbase == qbase ? -mb : PEN_X);
                    if (v < cm) { cm = v; dm = pm == best ? 0u : (pd == best ? 1u : 2u); }
                }
            }
            if (a > 0) {
                const int32_t pm = um, pd = ud, pi = ui;
                int32_t v = min(pm, pi) + PEN_O + PEN_E;
                if (pd + PEN_E < v) v = pd + PEN_E;
                if (v < INF32) { cd = v; dd = (pd + PEN_E == v) ? 1u : (pm <= pi ? 0u : 2u); }
            }
            if (b > 0) {
                const int32_t pm = lm, pd = ld, pi = li;
                int32_t v = min(pm, pd) + PEN_O + PEN_E;
                if (pi + PEN_E < v) v = pi + PEN_E;
                if (v < INF32) { ci = v; di = (pi + PEN_E == v) ? 2u : (pm <= pd ? 0u : 1u); }
            }
            cur[(b * 3) * LANE_STRIDE] = cm; cur[(b * 3 + 1) * LANE_STRIDE] = cd; cur[(b * 3 + 2) * LANE_STRIDE] = ci;
            sc.dirs[static_cast<size_t>(a * W + b) * LANE_STRIDE] = static_cast<uint8_t>(dm | (dd << 2) | (di << 4));
            lm = cm; ld = cd; li = ci;
            gm = um; gd = ud; gi = ui;
        }
        sc.lastcol[(a * 3) * LANE_STRIDE] = lm; sc.lastcol[(a * 3 + 1) * LANE_STRIDE] = ld; sc.lastcol[(a * 3 + 2) * LANE_STRIDE] = li;
        int32_t* t = prev; prev = cur; cur = t;
    }
    const int32_t* lastrow = prev;                                           // row n
    uint32_t ea = n, eb = m;
    int32_t best = INF32, em = INF32, ed = INF32, ei = INF32;
    if (mode == 2) {                                                         // the alignment may stop on the last row or column
        for (uint32_t b = 0; b <= m; b++) {
            const int32_t v = min(lastrow[(b * 3) * LANE_STRIDE], min(lastrow[(b * 3 + 1) * LANE_STRIDE], lastrow[(b * 3 + 2) * LANE_STRIDE]));
            if (v < best) { best = v; ea = n; eb = b; em = lastrow[(b * 3) * LANE_STRIDE]; ed = lastrow[(b * 3 + 1) * LANE_STRIDE]; ei = lastrow[(b * 3 + 2) * LANE_STRIDE]; }
        }
        for (uint32_t a = 0; a <= n; a++) {
            const int32_t v = min(sc.lastcol[(a * 3) * LANE_STRIDE], min(sc.lastcol[(a * 3 + 1) * LANE_STRIDE], sc.lastcol[(a * 3 + 2) * LANE_STRIDE]));
            if (v < best) { best = v; ea = a; eb = m; em = sc.lastcol[(a * 3) * LANE_STRIDE]; ed = sc.lastcol[(a * 3 + 1) * LANE_STRIDE]; ei = sc.lastcol[(a * 3 + 2) * LANE_STRIDE]; }
        }
    } else {
        em = lastrow[(m * 3) * LANE_STRIDE]; ed = lastrow[(m * 3 + 1) * LANE_STRIDE]; ei = lastrow[(m * 3 + 2) * LANE_STRIDE];
        best = min(em, min(ed, ei));
    }
    if (best >= INF32 || best > MAX_STEPS) return DP_DROPPED;
    uint32_t k = 0;
    if (mode == 2) {                                                         // the skipped suffix, as the aligner reports it
        for (uint32_t b = m; b > eb; b--) sc.ops[static_cast<size_t>(k++) * LANE_STRIDE] = 'I';
        for (uint32_t a = n; a > ea; a--) sc.ops[static_cast<size_t>(k++) * LANE_STRIDE] = 'D';
    }
    uint32_t a = ea, b = eb;
    uint32_t st = (em <= ed && em <= ei) ? 0u : (ed <= ei ? 1u : 2u);
    while (a > 0 || b > 0) {
        if (mode == 1 && (a == 0 || b == 0) && st == 0) break;               // reached the free border
        const uint32_t d = sc.dirs[static_cast<size_t>(a * W + b) * LANE_STRIDE];
        if (st == 0) {
            sc.ops[static_cast<size_t>(k++) * LANE_STRIDE] = S.r(i1 + a - 1) == S.q(j1 + b - 1) ? '=' : 'X';
            st = d & 3u; a--; b--;
        } else if (st == 1) {
            sc.ops[static_cast<size_t>(k++) * LANE_STRIDE] = 'D';
            const uint32_t dd = (d >> 2) & 3u;
            st = dd == 1 ? 1u : (dd == 0 ? 0u : 2u);
            a--;
        } else {
            sc.ops[static_cast<size_t>(k++) * LANE_STRIDE] = 'I';
            const uint32_t di = (d >> 4) & 3u;
            st = di == 2 ? 2u : (di == 0 ? 0u : 1u);
            b--;
        }
    }
    if (mode == 1) {                                                         // the skipped prefix
        for (; b > 0; b--) sc.ops[static_cast<size_t>(k++) * LANE_STRIDE] = 'I';
        for (; a > 0; a--) sc.ops[static_cast<size_t>(k++) * LANE_STRIDE] = 'D';
    }
    *n_ops = k;
    return best;
}

__device__ __forceinline__ uint32_t op_from_char(uint8_t ch) { return ch == '=' ? OP_EQ : ch == 'X' ? OP_X : ch == 'I' ? OP_I : OP_D; }

// dp_align for the stretches the rule leaves between two anchors of a long read — a handful of bases on either side — end to end and
// without a match bonus (the global aligner). Same recurrence, same tie rules, same walk back as dp_align above, but nothing goes
// through memory: the previous row of the three matrices lives in registers (the column loop is unrolled, a lane's columns beyond its m
// are skipped), a row of direction bytes is one 64-bit word, the rows are kept in eight of them (insert / extract by select), the bases
// of both stretches are packed into a register each (all their loads in flight together), and the operations leave as 2-bit codes
// (0 '=', 1 'X', 2 'I', 3 'D'; operation t of the REVERSED alignment in bits 2t..2t+1 of *ops2). dp_align keeps rows, directions and
// operations in the lane's scratch: seven memory operations per cell, each cell behind the stores of the one before it.
// (Tried: values times four + rank so that "smallest of three, ties in order, and which" is one v_min3_u32 — half the vector
// instructions per cell, 2.5 ms MORE for the launch: the kernel does not wait for this arithmetic.)
constexpr uint32_t DP_SMALL = 7;                                             // both stretches at most this long (and at least one base)
__device__ inline int dp_align_small(const Seqs& S, uint32_t i1, uint32_t n, uint32_t j1, uint32_t m, Scratch& sc, uint32_t* n_ops, uint32_t* ops2) {
    constexpr uint32_t W8 = DP_SMALL + 1;
    sc.cells += static_cast<unsigned long long>(n + 1) * (m + 1);
    // an index beyond the stretch reads the stretch's last base again, unused
    uint64_t qpack = 0, rpack = 0;
    for (uint32_t b = 0; b < m; b++) qpack |= static_cast<uint64_t>(S.q(j1 + b)) << (8 * b);
    for (uint32_t a = 0; a < n; a++) rpack |= static_cast<uint64_t>(S.r(i1 + a)) << (8 * a);
    int32_t pm[W8], pd[W8], pi[W8];                                          // row a - 1 (then row a, column by column)
    uint64_t drow[W8];
#pragma unroll
    for (uint32_t b = 0; b < W8; b++) { pm[b] = pd[b] = pi[b] = INF32; drow[b] = 0; }
    for (uint32_t a = 0; a <= n; a++) {
        const uint32_t rbase = a > 0 ? static_cast<uint32_t>(rpack >> (8 * (a - 1))) & 0xFFu : 0u;
        int32_t lm = INF32, ld = INF32, li = INF32;                            // cell (a, b - 1)
        int32_t gm = INF32, gd = INF32, gi = INF32;                            // cell (a - 1, b - 1)
        uint64_t row = 0;
#pragma unroll
        for (uint32_t b = 0; b < W8; b++) {
            if (b <= m) {
                int32_t cm = INF32, cd = INF32, ci = INF32;
                uint32_t dm = 3, dd = 0, di = 0;
                const int32_t um = a > 0 ? pm[b] : INF32, ud = a > 0 ? pd[b] : INF32, ui = a > 0 ? pi[b] : INF32;      // cell (a - 1, b)
                if (a == 0 && b == 0) cm = 0;
                if (a > 0 && b > 0) {
                    const int32_t best = min(gm, min(gd, gi));
                    if (best < INF32) {
                        const uint32_t qbase = static_cast<uint32_t>(qpack >> (8 * (b - 1))) & 0xFFu;
                        const int32_t v = best + (rbase == qbase ? 0 : PEN_X);
                        if (v < cm) { cm = v; dm = gm == best ? 0u : (gd == best ? 1u : 2u); }
                    }
                }
                if (a > 0) {
                    int32_t v = min(um, ui) + PEN_O + PEN_E;
                    if (ud + PEN_E < v) v = ud + PEN_E;
                    if (v < INF32) { cd = v; dd = (ud + PEN_E == v) ? 1u : (um <= ui ? 0u : 2u); }
                }
                if (b > 0) {
                    int32_t v = min(lm, ld) + PEN_O + PEN_E;
                    if (li + PEN_E < v) v = li + PEN_E;
                    if (v < INF32) { ci = v; di = (li + PEN_E == v) ? 2u : (lm <= ld ? 0u : 1u); }
                }
                pm[b] = cm; pd[b] = cd; pi[b] = ci;
                row |= static_cast<uint64_t>(dm | (dd << 2) | (di << 4)) << (8 * b);
                lm = cm; ld = cd; li = ci;
                gm = um; gd = ud; gi = ui;
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < W8; k++) drow[k] = a == k ? row : drow[k];
    }
    int32_t em = INF32, ed = INF32, ei = INF32;
#pragma unroll
    for (uint32_t b = 0; b < W8; b++) if (b == m) { em = pm[b]; ed = pd[b]; ei = pi[b]; }
    const int32_t best = min(em, min(ed, ei));
    if (best >= INF32 || best > MAX_STEPS) return DP_DROPPED;
    uint32_t k = 0, ops = 0;
    uint32_t a = n, b = m;
    uint32_t st = (em <= ed && em <= ei) ? 0u : (ed <= ei ? 1u : 2u);
    while (a > 0 || b > 0) {
        uint64_t row = 0;
#pragma unroll
        for (uint32_t r = 0; r < W8; r++) row = a == r ? drow[r] : row;
        const uint32_t d = static_cast<uint32_t>(row >> (8 * b)) & 0xFFu;
        if (st == 0) {
            const bool eq = (static_cast<uint32_t>(rpack >> (8 * (a - 1))) & 0xFFu) == (static_cast<uint32_t>(qpack >> (8 * (b - 1))) & 0xFFu);
            ops |= (eq ? 0u : 1u) << (2 * k); k++;
            st = d & 3u; a--; b--;
        } else if (st == 1) {
            ops |= 3u << (2 * k); k++;
            const uint32_t dd = (d >> 2) & 3u;
            st = dd == 1 ? 1u : (dd == 0 ? 0u : 2u);
            a--;
        } else {
            ops |= 2u << (2 * k); k++;
            const uint32_t di = (d >> 4) & 3u;
            st = di == 2 ? 2u : (di == 0 ? 0u : 1u);
            b--;
        }
    }
    *n_ops = k; *ops2 = ops;
    return best;
}

// Aligner::align::<LEFT_CLIPPING> (wfa.rs:254-299). semiglobal: 0 global aligner, 1 LEFT, 2 RIGHT free ends
// SMALL: short end-to-end stretches take dp_align_small (the kernel's build for long CIGARs; in the one for short reads, which has a
// quarter fewer registers per lane, its arrays would spill)
template <bool SMALL>
__device__ inline int aligner_align(const Seqs& S, uint32_t i1, uint32_t n, uint32_t j1, uint32_t m, int semiglobal, bool left_clipping, DCigar& cg,
                                    Scratch& sc) {
    uint32_t n_ops = 0;
    // (the same acceptance as dp_align's: a level whose scratch would refuse the stretch refuses it here too)
    if (SMALL && semiglobal == 0 && !left_clipping && n >= 1 && m >= 1 && n <= DP_SMALL && m <= DP_SMALL && n <= sc.lim.dp_dim && m <= sc.lim.dp_dim && (n + 1) * (m + 1) <= sc.lim.dp_cells) {
        uint32_t ops2 = 0;
        const int pen = dp_align_small(S, i1, n, j1, m, sc, &n_ops, &ops2);
        if (pen == DP_DROPPED) return align_simple(S, i1, n, j1, m, cg);
        for (uint32_t t = n_ops; t-- > 0;) {
            const uint32_t c = (ops2 >> (2 * t)) & 3u;
            cg.push_checked(c == 0 ? OP_EQ : c == 1 ? OP_X : c == 2 ? OP_I : OP_D, 1);
        }
        return -pen;
    }
    const int pen = dp_align(S, i1, n, j1, m, semiglobal ? max(1, PEN_X / 2) : 0, semiglobal, sc, &n_ops);
    // DP_TOO_BIG: a stand-in that consumes both stretches completely, as any end-to-end alignment does — the caller looks at sc.big
    if (pen == DP_DROPPED || pen == DP_TOO_BIG) return align_simple(S, i1, n, j1, m, cg);
    bool no_matches_yet = true;
    for (uint32_t t = n_ops; t-- > 0;) {                                     // sc.ops is reversed
        const uint32_t op = op_from_char(sc.ops[static_cast<size_t>(t) * LANE_STRIDE]);
        if (left_clipping && no_matches_yet && op == OP_EQ) {
            no_matches_yet = false;
            const uint32_t soft = cg.qlen;
            cg.clear();
            if (soft > 0) cg.push_unchecked(OP_I, soft);
        }
        cg.push_checked(op, 1);
    }
    if (left_clipping && no_matches_yet) {
        const uint32_t soft = cg.qlen;
        cg.clear();
        if (soft > 0) cg.push_unchecked(OP_I, soft);
    }
    return -pen;
}

// smart_align (wfa.rs:301-347); max_gap 0xFFFFFFFF = the `()` threshold
template <bool SMALL>
__device__ inline int smart_align(const Seqs& S, uint32_t i1, uint32_t i2, uint32_t j1, uint32_t j2, uint32_t max_gap, DCigar& cg, Scratch& sc) {
    const uint32_t jump1 = i2 - i1, jump2 = j2 - j1;
    if (jump1 > 0 && jump2 > 0) {
        const uint32_t safe_mismatch = (2 * PEN_O + 2 * PEN_E) / PEN_X;      // wfa.rs:212
        if (max_gap < jump1 || max_gap < jump2) return align_simple(S, i1, jump1, j1, jump2, cg);
        if (jump1 == jump2 && jump1 <= safe_mismatch) {
            int ndiff = 0;
            for (uint32_t t = 0; t < jump1; t++) {
                const bool eq = S.r(i1 + t) == S.q(j1 + t);
                cg.push_checked(eq ? OP_EQ : OP_X, 1);
                ndiff -= !eq;
            }
            return ndiff * PEN_X;
        }
        return aligner_align<SMALL>(S, i1, jump1, j1, jump2, 0, false, cg, sc);
    }
    if (jump1 > 0) { cg.push_unchecked(OP_D, jump1); return -PEN_O - static_cast<int>(jump1) * PEN_E; }
    if (jump2 > 0) { cg.push_unchecked(OP_I, jump2); return -PEN_O - static_cast<int>(jump2) * PEN_E; }
    return 0;
}

struct Job { uint32_t i1, n, j1, m; int semiglobal; bool left_clipping; };
// Cigar::optimize (cigar.rs:1167-1237). As upstream, the reference positions are counted from the start of the CIGAR while the
// sequence handed in is the whole target haplotype (cigar.rs:1362-1364): kept as written.
// Resumable, like the walk: a pass over the items that copies them into the free CIGAR buffer and, for every stretch between two
// anchors that holds both an insertion and a deletion, wants smart_align. Called in place, each lane ran its stretches alone while
// the other 63 waited — 40 of the kernel's 100 ms on 10-kb reads; now opt_step hands the stretch out (true, `job` = reference
// start / length, query start / length) and the caller runs smart_align for all lanes that hold one, at one call site.
struct OptState {
    DCigar nc;
    ItemWindow win;        // items of `self` from the last anchor on
    uint32_t i, j, qpos1, rpos1, qpos2, rpos2, flag, stage;
    bool have;
};
__device__ inline void opt_init(OptState& o, DCigar& self, Scratch& sc, uint32_t* slice_a, uint32_t* slice_b) {
    self.finish();
    o.win.init(self);
    if (self.keep) o.win.fill(self, 0);                                       // (a CIGAR of one chunk: the window is its slice, nothing is loaded)
    o.nc.init(sc.cig_free, sc.lim.cigar_cap, self.slice == slice_a ? slice_b : slice_a);      // the new CIGAR collects in the other slice
    o.i = o.j = 0; o.qpos1 = o.rpos1 = o.qpos2 = o.rpos2 = 0; o.flag = 0; o.stage = 0; o.have = false;
}
__device__ __forceinline__ void opt_begin_copy(OptState& o, const DCigar& self) {
    if (!o.have) { o.have = true; copy_items(self, 0, o.i, o.nc); o.nc.qlen = o.qpos1; o.nc.rlen = o.rpos1; }
}
__device__ __forceinline__ void opt_past_anchor(OptState& o, const DCigar& self, uint32_t op, uint32_t len) {
    o.qpos2 += len; o.rpos2 += len; o.qpos1 = o.qpos2; o.rpos1 = o.rpos2; o.flag = 0;
    if (o.have) {
        for (uint32_t k = o.i; k < o.j; k++) o.nc.push_raw(o.win.item(self, k));
        o.nc.push_checked(op, len);
        o.nc.qlen = o.qpos2; o.nc.rlen = o.rpos2;
    }
    o.i = o.j + 1;
    o.j++;
}
// true: align reference [job.i1, +job.n) with query [job.j1, +job.m) into o.nc (smart_align without a maximum gap), then call again;
// false: the pass is complete, or (opt_wants_items) it stands at an item its window does not hold — fill, then call again
__device__ __forceinline__ bool opt_wants_items(const OptState& o, const DCigar& self) { return o.stage == 0 && o.j < self.n && !o.win.has(o.j); }
__device__ __forceinline__ void opt_fill(OptState& o, const DCigar& self) {
    // from the last anchor, so that the items a stretch copies are in the window too — unless the stretch is as long as the window
    o.win.fill(self, o.j - o.i < ITEM_WIN - 4 ? o.i : o.j);
}
__device__ inline bool opt_step(OptState& o, DCigar& self, uint32_t max_gap, uint32_t anchor_size, Job& job) {
    for (;;) {
        if (o.stage == 0) {                                                   // over the items
            if (o.j >= self.n) { o.stage = 2; continue; }
            if (!o.win.has(o.j)) return false;                                // the caller fills the windows of all lanes (opt_wants_items)
            const uint2 item_j = o.win.get(o.j);
            const uint32_t op = item_j.x, len = item_j.y;
            const bool cq = cons_q(op), cr = cons_r(op);
            if (cq && cr && len >= anchor_size) {
                const uint32_t qshift = o.qpos2 - o.qpos1, rshift = o.rpos2 - o.rpos1;
                if (o.flag == 3 && !(max_gap < qshift) && !(max_gap < rshift)) {
                    opt_begin_copy(o, self);
                    job = Job{o.rpos1, o.rpos2 - o.rpos1, o.qpos1, o.qpos2 - o.qpos1, 0, false};
                    o.stage = 1;
                    return true;
                }
                opt_past_anchor(o, self, op, len);
            } else {
                o.qpos2 += cq ? len : 0; o.rpos2 += cr ? len : 0;
                o.flag |= (cq ? 0u : 1u) | ((cr ? 0u : 1u) << 1);
                o.j++;
            }
        } else if (o.stage == 1) {                                            // back from the aligner, in front of the anchor item j
            o.i = o.j;
            const uint2 item_j = o.win.item(self, o.j);
            opt_past_anchor(o, self, item_j.x, item_j.y);
            o.stage = 0;
        } else if (o.stage == 2) {                                            // what is behind the last anchor
            const uint32_t qshift = o.qpos2 - o.qpos1, rshift = o.rpos2 - o.rpos1;
            o.stage = 4;
            if (o.flag == 3 && !(max_gap < qshift) && !(max_gap < rshift)) {
                opt_begin_copy(o, self);
                job = Job{o.rpos1, o.rpos2 - o.rpos1, o.qpos1, o.qpos2 - o.qpos1, 0, false};
                o.stage = 3;
                return true;
            }
        } else if (o.stage == 3) {
            o.i = self.n;
            o.stage = 4;
        } else {
            if (o.have) {
                copy_items(self, o.i, self.n, o.nc);
                o.nc.finish();
                // self.tuples = new_cigar.tuples (lengths stay): the new items stay where they are, `self` looks there from now on
                self.t = o.nc.t; self.wb = nullptr; self.keep = o.nc.keep; self.slice = o.nc.slice;
                self.n = o.nc.n;
                self.overflow |= o.nc.overflow;
            }
            return false;
        }
    }
}

// double_cigar_move_and_shift (cigar.rs:1422-1466)
__device__ __forceinline__ uint32_t cons_class(uint32_t op) { return cons_q(op) && cons_r(op) ? 0u : (cons_q(op) ? 1u : 2u); }
__device__ inline uint32_t double_move(uint32_t op1, uint32_t op2, uint32_t& pos1, uint32_t& rem1, uint32_t& pos2, uint32_t& rem2) {
    // bit 0 read moves, 1 read CIGAR shifts, 2 haplotype moves, 3 haplotype CIGAR shifts; index = class(op1) * 3 + class(op2)
    // the nine cases as nibbles of one constant, case 0 lowest: {0xF, 0xB, 0xC, 0x3, 0x3, 0xF, 0xE, 0xA, 0xC}
    const uint32_t f = static_cast<uint32_t>(0xCAEF33CBFull >> (4u * (cons_class(op1) * 3 + cons_class(op2)))) & 0xFu;
    const bool rs = f & 2u, hs = f & 8u;
    const uint32_t shift = (rs && (!hs || rem1 <= rem2)) ? rem1 : rem2;
    pos1 += (f & 1u) ? shift : 0; rem1 -= rs ? shift : 0;
    pos2 += (f & 4u) ? shift : 0; rem2 -= hs ? shift : 0;
    return shift;
}

// the read's own CIGAR as it is stored: raw BAM words, hard clips at the ends count as soft ones (cigar.rs:309-320)
constexpr uint32_t SRC_LDS_WORDS = 1024;
struct SrcCigar {
    const uint32_t* raw; uint32_t n; bool hard_to_soft;
    const uint32_t* lds; uint32_t lds_n;                   // the first lds_n words once more, in LDS (shared by the lanes)
    __device__ __forceinline__ uint2 item(uint32_t i) const {
        const uint32_t w = i < lds_n ? lds[i] : raw[i];
        uint32_t op = w & 15u;
        if (hard_to_soft && op == OP_H && (i == 0 || i + 1 == n)) op = OP_S;
        return make_uint2(op, w >> 4);
    }
    __device__ uint32_t ref_len() const {
        uint32_t r = 0;
        for (uint32_t i = 0; i < n; i++) { const uint2 it = item(i); if (cons_r(it.x)) r += it.y; }
        return r;
    }
};

// Cigar::transfer_alignment::<false> as called by transfer_read_alignment (cigar.rs:1248-1384): anchor size 5, no maximum gap.
// jk: items of the haplotype-to-haplotype CIGAR (query = lower contig id), dir_jk 0 = QueryToRef, 1 = RefToQuery.
//
// The walk is written as a resumable state machine: the lanes of a wavefront carry different transfers and reach their aligner calls
// at different points of their walks; if the dynamic programme were called from inside the walk, the wavefront would run it once
// per distinct call point with one lane active. Instead a lane walks until it needs the aligner, hands the stretch out as a `Job`
// and waits; the caller runs the aligner for all waiting lanes at one converged call site and resumes the walks.
struct Walk {
    const uint2* jk_items; uint32_t jk_n; int dir_jk;
    uint32_t jk, op2, len2, rem2;
    uint32_t ijx, op1, len1, rem1;
    uint32_t last1, pos1, last2, pos2, start_k;
    int add;
    uint32_t phase;
    uint32_t n_jobs;           // stretches left for the aligner so far: markers {JOB_MARK, index} in the CIGAR under construction
    uint2 nxt1, nxt2;          // the items the two CIGARs continue with (ij item ijx, jk item jk), requested a step before they are needed
};
enum : uint32_t { PH_TOP = 0, PH_POST_LEFT, PH_POST, PH_TAIL, PH_POST_RIGHT, PH_FINISH, PH_DONE };

// returns false when the whole read alignment is a copy (the read lies inside one long match of the two haplotypes): `out` is final
__device__ inline bool walk_init(Walk& w, const uint2* jk_items, uint32_t jk_n, int dir_jk, uint32_t start_j, uint32_t off_ix, uint32_t off_qpos,
                                 uint32_t off_rpos, const SrcCigar& ij, DCigar& out) {
    const uint32_t FULL_MATCH_PADDING = 3;
    out.clear();
    w.jk_items = jk_items; w.jk_n = jk_n; w.dir_jk = dir_jk;
    w.jk = off_ix;
    w.op2 = dir_jk ? op_invert(jk_items[w.jk].x) : jk_items[w.jk].x;
    const uint32_t init_shift = start_j - off_qpos;
    w.len2 = jk_items[w.jk].y; w.rem2 = w.len2 - init_shift;
    w.jk++;
    w.start_k = off_rpos + (cons_r(w.op2) ? init_shift : 0);
    if (w.op2 == OP_EQ && init_shift >= FULL_MATCH_PADDING && w.rem2 >= ij.ref_len() + FULL_MATCH_PADDING) {
        for (uint32_t i = 0; i < ij.n; i++) { const uint2 it = ij.item(i); out.push_unchecked(it.x, it.y); }
        out.flush();
        w.phase = PH_DONE;
        return false;
    }
    const uint2 it1 = ij.item(0);
    w.len1 = it1.y; w.rem1 = w.len1; w.op1 = it1.x;
    w.ijx = 1;
    w.nxt1 = w.ijx < ij.n ? ij.item(w.ijx) : make_uint2(0, 0);
    w.nxt2 = w.jk < w.jk_n ? jk_items[w.jk] : make_uint2(0, 0);
    w.last1 = 0; w.pos1 = 0; w.last2 = w.start_k; w.pos2 = w.start_k;
    w.add = -1;
    w.n_jobs = 0;
    w.phase = PH_TOP;
    return true;
}

// walks on until the aligner is needed (WALK_JOB, `job` filled), the transfer is complete (WALK_DONE) or `budget` phases have been
// walked (WALK_MORE; a knob in round 3, a constant now: no budget). Measured at 10-kb reads x 256 alleles: letting the lanes that need the
// aligner wait for ALL others to need it too (no budget) is the fastest form — 231 ms against 350 ms when the wavefront looks after
// every phase: a call of the aligner costs the wavefront the same whatever the number of lanes in it, so few full calls beat many
// sparse ones, although the lanes then spend three quarters of the walk waiting (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = 16 of 64).
enum : uint32_t { WALK_DONE = 0, WALK_JOB = 1, WALK_MORE = 2, WALK_ASSEMBLE = 3, WALK_OPTIMIZE = 4 };
constexpr uint32_t JOB_MARK = 15;                           // operation code of a marker item: its length is the index of the stretch
__device__ inline uint32_t walk_step(Walk& w, const SrcCigar& ij, const Seqs& S, DCigar& out, Scratch& sc, Job& job, uint32_t budget) {
    const uint32_t anchor_size = 5, ANCHOR_MARGIN = 5, CLIP_PADDING = 3;
    const uint32_t len_i = S.read_len, len_k = S.target_len;
    for (;; budget--) {
        if (budget == 0) return WALK_MORE;
        if (w.phase == PH_TOP) {
            int add = -1;
            const bool e1 = w.op1 == OP_EQ, e2 = w.op2 == OP_EQ;
            if (e1 && e2) { if (min(w.rem1, w.rem2) >= anchor_size) add = OP_EQ; }
            else if (e1 && !e2) { if (w.rem1 >= anchor_size && w.len1 - w.rem1 >= ANCHOR_MARGIN) add = static_cast<int>(w.op2); }
            else if (!e1 && e2) { if (w.rem2 >= anchor_size && w.len2 - w.rem2 >= ANCHOR_MARGIN) add = static_cast<int>(w.op1); }
            w.add = add;
            w.phase = PH_POST;
            if (add >= 0) {
                if (w.last1 == 0 && w.pos1 > 0) {
                    // align_ends::<LEFT> (wfa.rs:349-365)
                    const uint32_t from = w.last2 > w.pos1 + CLIP_PADDING ? w.last2 - (w.pos1 + CLIP_PADDING) : 0;     // saturating_sub
                    w.phase = PH_POST_LEFT;
                    if (from == w.pos2) out.push_unchecked(OP_I, w.pos1 - w.last1);
                    else { job = Job{from, w.pos2 - from, w.last1, w.pos1 - w.last1, 1, true}; return WALK_JOB; }
                } else {
                    // smart_align (wfa.rs:301-347) without a maximum gap; only the dynamic programme is handed out
                    const uint32_t jump1 = w.pos2 - w.last2, jump2 = w.pos1 - w.last1;
                    if (jump1 > 0 && jump2 > 0) {
                        constexpr uint32_t safe_mismatch = (2 * PEN_O + 2 * PEN_E) / PEN_X;  // wfa.rs:212
                        if (jump1 == jump2 && jump1 <= safe_mismatch) {
                            // (one base: the short-read case. Else the bases of the whole stretch first — at most three of each — then the
                            // pushes: their stores would hold the loads up)
                            if (jump1 == 1) out.push_checked(S.r(w.last2) == S.q(w.last1) ? OP_EQ : OP_X, 1);
                            else {
                            bool eq[safe_mismatch];
#pragma unroll
                            for (uint32_t t = 0; t < safe_mismatch; t++) eq[t] = S.r(w.last2 + min(t, jump1 - 1)) == S.q(w.last1 + min(t, jump1 - 1));
#pragma unroll
                            for (uint32_t t = 0; t < safe_mismatch; t++) if (t < jump1) out.push_checked(eq[t] ? OP_EQ : OP_X, 1);
                            }
                        } else {
                            // A stretch for the aligner between two anchors. Nothing of the walk depends on how it aligns (an end-to-end
                            // alignment consumes both stretches completely, whatever its operations): the stretch is noted, a marker takes
                            // its place in the CIGAR and the walk goes on. The aligner runs when the walk has reached the end of the
                            // read's CIGAR (assemble_jobs, WALK_ASSEMBLE) — for all lanes of the wavefront together, stretch by stretch.
                            // Calling it from here made every lane wait at every stretch of every other lane: an indel every fifty
                            // bases of a long read, at another step in every lane (16 of 64 lanes active, profiles/r03_pmc_transfer_*).
                            if (w.n_jobs < sc.lim.cigar_cap / 2 + 8) {
                                sc.jobs[static_cast<size_t>(w.n_jobs) * LANE_STRIDE] = make_uint4(w.last2, jump1, w.last1, jump2);
                                out.push_raw(make_uint2(JOB_MARK, w.n_jobs));
                                out.rlen += jump1; out.qlen += jump2;
                                w.n_jobs++;
                            } else out.overflow = true;
                        }
                    } else if (jump1 > 0) out.push_unchecked(OP_D, jump1);
                    else if (jump2 > 0) out.push_unchecked(OP_I, jump2);
                }
            }
        }
        if (w.phase == PH_POST_LEFT) {
            w.start_k = w.start_k + w.pos2 - w.last2 - out.rlen;
            w.phase = PH_POST;
        }
        if (w.phase == PH_POST) {
            const uint32_t shift = double_move(w.op1, w.op2, w.pos1, w.rem1, w.pos2, w.rem2);
            if (w.add >= 0) { out.push_checked(static_cast<uint32_t>(w.add), shift); w.last1 = w.pos1; w.last2 = w.pos2; }
            w.phase = PH_TOP;
            if (w.rem1 == 0) {
                if (w.ijx == ij.n) w.phase = PH_TAIL;
                else {
                    const uint2 it1 = w.nxt1; w.len1 = it1.y; w.rem1 = w.len1; w.op1 = it1.x; w.ijx++;
                    if (w.ijx < ij.n) w.nxt1 = ij.item(w.ijx);                   // used a step from now at the earliest
                }
            }
            if (w.phase == PH_TOP && w.rem2 == 0) {
                if (w.jk == w.jk_n) w.phase = PH_TAIL;
                else {
                    const uint2 it2 = w.nxt2; w.len2 = it2.y; w.rem2 = w.len2; w.op2 = w.dir_jk ? op_invert(it2.x) : it2.x; w.jk++;
                    if (w.jk < w.jk_n) w.nxt2 = w.jk_items[w.jk];
                }
            }
        }
        if (w.phase < PH_TAIL) continue;                                         // the common round: one step of the two CIGARs
        if (w.phase == PH_TAIL) {
            if (w.n_jobs) return WALK_ASSEMBLE;                                  // the caller resolves the markers (assemble_jobs), then comes back
            w.phase = PH_FINISH;
            if (w.last1 != len_i) {
                // align_ends::<RIGHT>
                const uint32_t i1 = w.last2, i2 = min(len_k, w.last2 + len_i - w.last1 + CLIP_PADDING);
                if (i1 == i2) out.push_unchecked(OP_I, len_i - w.last1);
                else { w.phase = PH_POST_RIGHT; job = Job{i1, i2 - i1, w.last1, len_i - w.last1, 2, false}; return WALK_JOB; }
            }
        }
        if (w.phase == PH_POST_RIGHT) {
            uint32_t soft = 0;
            out.flush();
            while (out.n && out.get(out.n - 1).x != OP_EQ) {                    // pop_if(op != Equal)
                const uint2 it = out.pop_back();
                if (cons_q(it.x)) { out.qlen -= it.y; soft += it.y; }
                if (cons_r(it.x)) out.rlen -= it.y;
            }
            if (soft > 0) out.push_unchecked(OP_I, soft);
            w.phase = PH_FINISH;
        }
        if (w.phase == PH_FINISH) return WALK_OPTIMIZE;                          // the caller: optimize_and_finish for all lanes together
        if (w.phase == PH_DONE) return WALK_DONE;
    }
}

// Cigar::optimize (MAX_OPTIMIZATION_GAP 20, OPTIMIZATION_ANCHOR 5) + boundary_ins_to_soft (cigar.rs:554-561) of the finished transfers
// of a wavefront: `mine` = this lane has one. The lanes scan their CIGARs; those that stand in front of a stretch call smart_align at one site.
// (diag: developer build, four counters — ticks inside smart_align, rounds, lanes with a stretch, largest matrix of a round)
template <bool SMALL>
__device__ inline void optimize_and_finish(bool mine, DCigar& out, const Seqs& S, Scratch& sc, uint32_t* slice_a, uint32_t* slice_b,
                                           unsigned long long* diag = nullptr) {
    OptState o;
    Job job;
    bool active = mine;
    if (mine) opt_init(o, out, sc, slice_a, slice_b);
    do {
        // a round: every lane that is scanning reads its next items (all lanes' loads in flight together) and scans on to a stretch, the end
        // of its window or the end of the CIGAR; the lanes that stand in front of a stretch call smart_align (see the assemble loop)
        bool need = false;
        if (active) {
            need = opt_step(o, out, 20, 5, job);
            if (!need && !opt_wants_items(o, out)) active = false;
        }
        if (active && !need) opt_fill(o, out);                                // at the end of its window (the first round: of the empty one)
        unsigned long long t0 = 0;
        if (diag) { __builtin_amdgcn_sched_barrier(0); t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
        if (need) smart_align<SMALL>(S, job.i1, job.i1 + job.n, job.j1, job.j1 + job.m, 0xFFFFFFFFu, o.nc, sc);
        if (diag) {
            __builtin_amdgcn_sched_barrier(0);
            diag[0] += __builtin_amdgcn_s_memtime() - t0; diag[1]++;
            diag[2] += static_cast<unsigned long long>(__popcll(__ballot(need)));
            uint32_t mx = need ? (job.n + 1) * (job.m + 1) : 0u;
            for (int k = 32; k > 0; k >>= 1) mx = max(mx, static_cast<uint32_t>(__shfl_xor(static_cast<int>(mx), k)));
            diag[3] += mx;
        }
    } while (__any(active));
    if (mine && out.n) {
        const uint2 first = out.get(0);
        if (first.x == OP_I) out.set(0, make_uint2(OP_S, first.y));
        const uint2 last = out.get(out.n - 1);
        if (last.x == OP_I) out.set(out.n - 1, make_uint2(OP_S, last.y));
    }
}

}  // namespace xfer
}  // namespace lcty
