// lcty_map_long.hip — candidate generation inside a locus, the LONG route (second slice): read ends of
// any length (HiFi / ONT reads of tens of kilobases) on up to 256 basis alleles. The reference hands this to minimap2
// (-x map-ont | map-hifi -N min(25 000, 4 x alleles) -f 0.05 --eqx, src/command/genotype.rs:990-1002) and reads its `aln.bam`; no
// source of that mapper is in the reference tree. The algorithm here is this build's own (seed - chain - align, as minimap2-class
// mappers do), stated below and restated in tests/pyref_map_long.py, against which the kernels are bit-exact.
//
// Coordinates of an (allele, strand) group: q = position on the read end AS SEQUENCED, t = position on the allele in the read's
// orientation (strand 0: the allele itself; strand 1: its reverse complement). Everything up to the record is done in (q, t); the
// record (position, CIGAR order) is turned into BAM orientation when it is written.
//
//   seeds    as on the short route (every `stride`-th k-mer plus the last one, k-mers with a base that is not ACGT or with more than
//            max_occ places in the index skipped), any number of them.
//   anchors  every (seed, place) pair is an anchor (q, t) of its group g = basis allele x 2 + strand; a group keeps the first
//            2 x seeds anchors, in seed order (within a seed: in index order).
//   chains   an anchor starts a chain (value k) or follows one of the last `chain_back` anchors j of its group: 0 < dq, dt <=
//            chain_gap, |dq - dt| <= chain_skew, and on another diagonal only k or more bases on in both sequences; value f(j) +
//            min(dq, dt, k) - (0 on the same diagonal, else 2 + |dq - dt|); the highest value wins, then the most recent anchor.
//            A group's chain ends at its anchor of the highest value (the first one).
//   align    the groups whose chain has >= min_votes anchors and >= half the value of the read end's best chain. Walking the chain
//            from its last anchor to its first: beyond the last anchor an extension (fixed start, free end: it ends on the aligned
//            base with the best total, end_bonus when the read end is reached, the rest is soft-clipped), between two anchors a piece
//            (fixed start and end; overlapping anchors of one diagonal are joined as they are), before the first anchor an extension
//            again (free start). Every segment is a gap-affine alignment over the nodes (i read bases, j allele bases taken) of a band of
//            diagonals j - i: -band .. +band for the extensions, min(0, d) - band .. max(0, d) + band for a piece whose corners are d
//            diagonals apart; H = best of (base step, deletion, insertion), gaps open from H at gap_open and go on at gap_extend a
//            base. Preference on ties: the base step before the deletion before the insertion, to go on before starting afresh,
//            to open before to extend, the first best end node by (i, diagonal).
//   records  as on the short route: best score first (the smallest (allele, strand) on ties), the others with score >= min_score
//            as secondary records, no candidate = an unmapped record; = / X / I / D / S CIGARs, SEQ in BAM orientation.
//
// Three kernels. map_long_chain_kernel, one wavefront per read end: lane = seed (index lookups), then lane = place of a seed (the
// chains of all groups grow side by side in a scratch of the workgroup; two places of one group in a batch take turns in index
// order), then lane = candidate: the chain of every group to be aligned is written out, last anchor first, with a work item.
// map_long_align_kernel, one wavefront per work item (taken from a cursor): lane = diagonal of the band. A row of nodes is computed
// at once — the base step and the insertion come from the row above (registers and a DPP shift for bands of up to 64 diagonals, two LDS
// rows in chunks of 64 lanes beyond), the deletions of the row are a prefix maximum over the lanes (a gap never opens from a deletion
// at a gain when gap_open >= gap_extend, which the long route asks for, so the openers are the nodes' other two states) — five
// direction bits per node stay in LDS (short segments) or go to a scratch of the wavefront, the walk back reads them in blocks through
// LDS, takes runs of equal base steps at once, and leaves the CIGAR runs right to left in scratch; they are copied into the arena at
// the end. Two kinds of piece never reach the rows: equal lengths with so few mismatches that the diagonal provably beats any gap, and
// sides that differ by one gap only. map_long_emit_kernel<WRITE>, one wavefront per read end, runs twice: sizes, then records (host
// prefix sums in between).
#include <algorithm>
#include <chrono>

#include "lcty_map_internal.hpp"

namespace lcty {

namespace {

constexpr int32_t LNEG = -(1 << 29);
constexpr uint32_t LONG_MAX_GROUPS = 2 * MAP_LONG_MAX_BASIS;

struct LongCand { int32_t score; uint32_t pos, ops_at, n_words, g, pad; };
struct LongWork { uint32_t mate, g_slot, chain_at, chain_n; };         // g | slot of the read end's candidates << 16

struct LongView {
    const MapSlot* table; uint64_t mask;
    const uint64_t* entries; const uint16_t* basis;
    uint32_t n_basis, k, stride, min_votes, max_occ, band, chain_gap, chain_skew, chain_back;
    int32_t match, mismatch, end_bonus, min_score, gap_open, gap_extend;
    const uint8_t* seqs; const uint64_t* seq_off; const uint32_t* allele_len;
    uint64_t n_mates;
    const uint32_t* mate_len; const uint64_t* mate_off; const uint32_t* bases2; const uint32_t* nmask;
    int paired;
    // kernel 1: scratch of a workgroup, and what it leaves
    uint4* anchors; uint32_t cap_g;               // [workgroup][group][cap_g]: q, t, value, back | anchors in the chain << 8
    uint2* chain; uint32_t chain_cap;             // the chains to be aligned: (q, t), last anchor first
    LongWork* work; uint32_t n_work;
    uint32_t* n_have;
    uint32_t* counters;                           // [0] work items, [1] CIGAR words asked for in `ops`, [2] the widest record, [3] chain entries asked for, [4] next work item of kernel 2
    // kernel 2: scratch of a wavefront, and what it leaves
    uint8_t* dirs; uint64_t dirs_bytes;           // [wavefront][dirs_bytes]
    uint32_t* opsbuf; uint32_t ops_wave;          // [wavefront][ops_wave]
    uint32_t wmax, tb_bytes;
    LongCand* cands; uint32_t slots;
    uint32_t* ops; uint32_t ops_cap;
    // kernel 3
    uint32_t* n_recs; uint32_t* n_cigar;
    const uint64_t* rec_at; const uint64_t* cig_at; const uint64_t* pair_cig;
    lcty_aln_rec* recs; uint32_t* cigar; uint32_t* out_bases2; uint32_t* out_nmask;
};

// ---- kernel 1: seeds -> anchors -> chains -> the chains to be aligned
__global__ __launch_bounds__(64) void map_long_chain_kernel(const LongView V) {
    __shared__ uint32_t g_n[LONG_MAX_GROUPS];
    __shared__ int32_t g_bf[LONG_MAX_GROUPS];
    __shared__ uint32_t g_bi[LONG_MAX_GROUPS];
    __shared__ uint32_t owner[LONG_MAX_GROUPS];
    __shared__ uint16_t cand_g[LONG_MAX_GROUPS];
    __shared__ uint32_t sd_start[64], sd_count[64], sd_pr[64], sd_fwd[64];
    const uint32_t lane = threadIdx.x;
    const uint32_t n_groups = 2 * V.n_basis;
    const uint32_t k = V.k;
    uint4* wg_anchors = V.anchors + static_cast<size_t>(blockIdx.x) * n_groups * V.cap_g;
    for (uint64_t m = blockIdx.x; m < V.n_mates; m += gridDim.x) {
        const uint32_t L = V.mate_len[m];
        if (L == 0) { if (lane == 0) V.n_have[m] = 0; continue; }          // absent read end
        const uint64_t off = V.mate_off[m];
        for (uint32_t g = lane; g < n_groups; g += 64) { g_n[g] = 0; g_bf[g] = LNEG; g_bi[g] = 0; owner[g] = 64; }
        __syncthreads();
        uint32_t n_seeds = 0, n0 = 0, span = 0;
        if (L >= k) { span = L - k; n0 = span / V.stride + 1; n_seeds = n0 + (span % V.stride ? 1u : 0u); }
        const uint32_t cap = min(2 * n_seeds, V.cap_g);
        for (uint32_t s0 = 0; s0 < n_seeds; s0 += 64) {
            const uint32_t sidx = s0 + lane;
            uint32_t start = 0, count = 0, pr = 0; bool read_fwd = false;
            if (sidx < n_seeds) {
                pr = sidx < n0 ? sidx * V.stride : span;
                uint64_t fw = 0, rv = 0; bool bad = false;
                for (uint32_t j = 0; j < k; j++) {
                    const uint32_t e = base_at(V.bases2, off, pr + j);
                    bad |= n_at(V.nmask, off, pr + j);
                    fw = (fw << 2) | e;
                    rv = (rv >> 2) | (static_cast<uint64_t>(3u - e) << (2 * k - 2));
                }
                if (!bad) {
                    read_fwd = fw <= rv;
                    const uint64_t canon = read_fwd ? fw : rv;
                    uint64_t h = map_hash(canon) & V.mask;
                    for (;;) {
                        const MapSlot sl = V.table[h];
                        if (sl.key == MAP_FREE) break;
                        if (sl.key == canon) { start = sl.start; count = sl.count > V.max_occ ? 0u : sl.count; break; }
                        h = (h + 1) & V.mask;
                    }
                }
            }
            sd_start[lane] = start; sd_count[lane] = count; sd_pr[lane] = pr; sd_fwd[lane] = read_fwd;
            __syncthreads();
            const uint32_t in_batch = min(64u, n_seeds - s0);
            for (uint32_t si = 0; si < in_batch; si++) {
                const uint32_t cnt = sd_count[si];
                if (cnt == 0) continue;
                const uint32_t st = sd_start[si], q = sd_pr[si]; const bool rfwd = sd_fwd[si] != 0;
                for (uint32_t e0 = 0; e0 < cnt; e0 += 64) {
                    bool pending = e0 + lane < cnt;
                    uint32_t g = 0, t = 0;
                    if (pending) {
                        const uint64_t en = V.entries[st + e0 + lane];
                        const uint32_t b = static_cast<uint32_t>(en >> 33), pa = static_cast<uint32_t>(en >> 1);
                        const uint32_t strand = rfwd == ((en & 1ull) != 0) ? 0u : 1u;
                        g = 2 * b + strand;
                        t = strand ? V.allele_len[V.basis[b]] - k - pa : pa;
                    }
                    // two places of one group take turns, in index order
                    while (__ballot(pending) != 0ull) {
                        if (pending) atomicMin(&owner[g], lane);
                        __syncthreads();
                        const bool mine = pending && owner[g] == lane;
                        if (mine) {
                            const uint32_t have = g_n[g];
                            if (have < cap) {
                                uint4* list = wg_anchors + static_cast<size_t>(g) * V.cap_g;
                                int32_t f = static_cast<int32_t>(k); uint32_t back = 0, cnt_chain = 1;
                                const uint32_t look = min(V.chain_back, have);
                                // eight anchors back at a time: the loads go out together, the anchors are looked at most recent first
                                for (uint32_t o0 = 1; o0 <= look; o0 += 8) {
                                    uint4 prev[8];
#pragma unroll
                                    for (uint32_t u = 0; u < 8; u++) prev[u] = list[o0 + u <= look ? have - o0 - u : have - look];
#pragma unroll
                                    for (uint32_t u = 0; u < 8; u++) {
                                        const uint4 a = prev[u];
                                        const uint32_t o = o0 + u;
                                        const int64_t dq = static_cast<int64_t>(q) - a.x, dt = static_cast<int64_t>(t) - a.y;
                                        const int64_t sk = dq > dt ? dq - dt : dt - dq;
                                        const bool ok = o <= look && dq > 0 && dt > 0 && dq <= V.chain_gap && dt <= V.chain_gap && sk <= V.chain_skew &&
                                                        (sk == 0 || (dq >= k && dt >= k));
                                        const int64_t gain = min(min(dq, dt), static_cast<int64_t>(k));
                                        const int32_t v = static_cast<int32_t>(a.z) + static_cast<int32_t>(gain) - (sk ? 2 + static_cast<int32_t>(sk) : 0);
                                        if (ok && v > f) { f = v; back = o; cnt_chain = (a.w >> 8) + 1; }
                                    }
                                }
                                list[have] = make_uint4(q, t, static_cast<uint32_t>(f), back | (cnt_chain << 8));
                                g_n[g] = have + 1;
                                if (have == 0 || f > g_bf[g]) { g_bf[g] = f; g_bi[g] = have; }
                            }
                        }
                        __syncthreads();
                        if (mine) { owner[g] = 64; pending = false; }
                        __syncthreads();
                    }
                }
            }
            __syncthreads();
        }
        // ---- the groups to be aligned, in (allele, strand) order
        int32_t top = LNEG;
        for (uint32_t g = lane; g < n_groups; g += 64) if (g_n[g]) top = max(top, g_bf[g]);
        for (int o = 32; o > 0; o >>= 1) top = max(top, __shfl_xor(top, o));
        uint32_t n_cand = 0;
        for (uint32_t g0 = 0; g0 < n_groups; g0 += 64) {
            const uint32_t g = g0 + lane;
            bool q = false;
            if (g < n_groups && g_n[g]) {
                const uint4 a = wg_anchors[static_cast<size_t>(g) * V.cap_g + g_bi[g]];
                q = 2 * static_cast<int64_t>(g_bf[g]) >= top && (a.w >> 8) >= V.min_votes;
            }
            const unsigned long long qm = __ballot(q);
            if (q) cand_g[n_cand + static_cast<uint32_t>(__popcll(qm & ((1ull << lane) - 1ull)))] = static_cast<uint16_t>(g);
            n_cand += static_cast<uint32_t>(__popcll(qm));
        }
        __syncthreads();
        // ---- lane = candidate: its chain, last anchor first, and a work item for kernel 2
        for (uint32_t c0 = 0; c0 < n_cand; c0 += 64) {
            const bool have = c0 + lane < n_cand;
            const uint32_t g = have ? cand_g[c0 + lane] : 0u;
            const uint4* list = wg_anchors + static_cast<size_t>(g) * V.cap_g;
            uint32_t at = have ? g_bi[g] : 0u;
            const uint32_t len = have ? list[at].w >> 8 : 0u;
            uint32_t incl = len;
            for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(incl, o); if (lane >= static_cast<uint32_t>(o)) incl += up; }
            const uint32_t total = static_cast<uint32_t>(__shfl(static_cast<int>(incl), 63));
            const uint32_t in_batch = min(64u, n_cand - c0);
            uint32_t cbase = 0, wbase = 0;
            if (lane == 0) { cbase = atomicAdd(&V.counters[3], total); wbase = atomicAdd(&V.counters[0], in_batch); }
            cbase = static_cast<uint32_t>(__shfl(static_cast<int>(cbase), 0)); wbase = static_cast<uint32_t>(__shfl(static_cast<int>(wbase), 0));
            const uint32_t mine_at = cbase + incl - len;
            if (have && cbase + total <= V.chain_cap && cbase + total >= cbase) {
                for (uint32_t idx = 0; idx < len; idx++) {
                    const uint4 a = list[at];
                    V.chain[mine_at + idx] = make_uint2(a.x, a.y);
                    at -= a.w & 0xFFu;
                }
                V.work[wbase + lane] = LongWork{static_cast<uint32_t>(m), g | ((c0 + lane) << 16), mine_at, len};
            }
        }
        if (lane == 0) V.n_have[m] = n_cand;
        __syncthreads();
    }
}

// ---- kernel 2: one wavefront aligns one group along its chain
// the read end of the work item and the group's allele in the read's orientation
struct Seqs {
    const uint32_t* b2; const uint32_t* nm; uint64_t off;
    const uint8_t* ref; uint32_t alen; uint32_t strand;
    __device__ __forceinline__ uint32_t read_base(uint32_t q) const { return n_at(nm, off, q) ? 4u : base_at(b2, off, q); }
    __device__ __forceinline__ uint32_t allele_base(uint32_t t) const {
        const uint32_t e = enc_of(strand ? ref[alen - 1 - t] : ref[t]);
        return e == 4u ? 4u : (strand ? 3u - e : e);
    }
    __device__ __forceinline__ bool eq(uint32_t q, uint32_t t) const {
        const uint32_t r = read_base(q);
        return r < 4u && r == allele_base(t);
    }
};

// the scratch of a wavefront and the CIGAR runs in the order they are met (right to left); every lane holds the same run state,
// lane 0 writes
struct WaveState {
    int32_t* hrow; int32_t* frow; uint8_t* tb;          // LDS: the running rows of a band, a block of direction bytes
    uint8_t* abuf; uint8_t* rbuf;                        // LDS: the allele bases under 64 rows of the band, the read bases of those rows
    uint8_t* dirs; uint32_t* ops;                        // scratch of the wavefront
    uint32_t tb_bytes;
    uint32_t n, cur_op, cur_len;
    __device__ __forceinline__ void put(uint32_t op, uint32_t len) {
        if (len == 0) return;
        if (op == cur_op) { cur_len += len; return; }
        flush();
        cur_op = op; cur_len = len;
    }
    __device__ __forceinline__ void flush() {
        if (cur_len) { if (threadIdx.x == 0) ops[n] = (cur_len << 4) | cur_op; n++; }
        cur_len = 0; cur_op = 0xFu;
    }
};

struct SegOut { int32_t score; uint32_t left, t_taken; };

// lane i takes lane i - 1, lane 0 takes `first` (DPP wave_shr:1)
__device__ __forceinline__ int32_t from_left(int32_t x, int32_t first) {
    return __builtin_amdgcn_update_dpp(first, x, 0x138, 0xF, 0xF, false);
}
// lane i takes lane i + 1, lane 63 takes `last` (DPP wave_shl:1)
__device__ __forceinline__ int32_t from_right(int32_t x, int32_t last) {
    return __builtin_amdgcn_update_dpp(last, x, 0x130, 0xF, 0xF, false);
}
// inclusive prefix maximum over the 64 lanes: inside the rows of 16 (row_shr 1, 2, 4, 8), then across them (row_bcast 15 / 31)
__device__ __forceinline__ int32_t prefix_max(int32_t v) {
    constexpr int32_t none = INT32_MIN;
    v = max(v, __builtin_amdgcn_update_dpp(none, v, 0x111, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(none, v, 0x112, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(none, v, 0x114, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(none, v, 0x118, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(none, v, 0x142, 0xA, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(none, v, 0x143, 0xC, 0xF, false));
    return v;
}

// One segment of an alignment along a chain (the header comment states the recurrence): read bases [q0, q0 + n), allele bases
// [t0, t0 + m). free_end: the extension beyond the last anchor; free_start: the one before the first anchor; neither: a piece.
// -> score, read bases left unaligned at the free side, allele bases taken. All arguments and results are the same in every lane.
__device__ SegOut wave_segment(const LongView& V, const Seqs& S, const uint32_t q0, const uint32_t n, const uint32_t t0, const uint32_t m,
                               const bool free_start, const bool free_end, WaveState& C) {
    const uint32_t lane = threadIdx.x;
    const int32_t B = static_cast<int32_t>(V.band);
    const int64_t d = static_cast<int64_t>(m) - static_cast<int64_t>(n);
    int64_t dlo, dhi;
    if (free_start) { dlo = d - B; dhi = d + B; }
    else if (free_end) { dlo = -B; dhi = B; }
    else { dlo = (d < 0 ? d : 0) - B; dhi = (d > 0 ? d : 0) + B; }
    const uint32_t W = static_cast<uint32_t>(dhi - dlo + 1);
    const uint32_t n_chunks = (W + 63) / 64;
    if (!free_start && !free_end && n == m && n > 0 && n <= C.tb_bytes * 8) {
        // A piece between corners on one diagonal whose bases differ in h places: any alignment with gaps has an insertion and a
        // deletion and at most n - 1 base steps, so it scores <= (n - 1) match - 2 gap_open; when the diagonal scores more than that
        // it is what the rows below would find (and the only one) — the lanes compare 64 bases at a time and the runs are written
        // from the masks. Nearly half the pieces of a 3 %-error read, nearly all of a HiFi read.
        uint64_t* masks = reinterpret_cast<uint64_t*>(C.tb);
        const uint32_t n_masks = (n + 63) / 64;
        uint32_t h = 0;
        __syncthreads();
        for (uint32_t c = 0; c < n_masks; c++) {
            const uint32_t x = c * 64 + lane;
            const bool same = x < n && S.eq(q0 + x, t0 + x);
            const unsigned long long sm = __ballot(same), in = __ballot(x < n);
            h += static_cast<uint32_t>(__popcll(in & ~sm));
            if (lane == 0) masks[c] = sm;
        }
        __syncthreads();
        if (static_cast<int64_t>(h) * (V.match + V.mismatch) < static_cast<int64_t>(V.match) + 2 * static_cast<int64_t>(V.gap_open)) {
            for (uint32_t c = n_masks; c-- > 0;) {
                const uint64_t bits = masks[c];
                uint32_t pos = min(64u, n - c * 64);                        // bases of this mask not written yet: [0, pos)
                while (pos > 0) {
                    const bool same = (bits >> (pos - 1)) & 1ull;
                    uint64_t other = same ? ~bits : bits;                   // set where the outcome differs from that of base pos - 1
                    if (pos < 64) other &= (1ull << pos) - 1ull;
                    const uint32_t run = other ? pos - (64u - static_cast<uint32_t>(__clzll(static_cast<long long>(other)))) : pos;
                    C.put(same ? 7u : 8u, run);
                    pos -= run;
                }
            }
            __syncthreads();
            return SegOut{static_cast<int32_t>((n - h) * V.match) - static_cast<int32_t>(h * V.mismatch), 0u, m};
        }
    }
    if (!free_start && !free_end && n != m && V.gap_open > V.gap_extend) {
        // A piece whose two sides differ by one gap only — the shorter side equals the longer one with |d| bases taken out: a common
        // prefix of p bases (on the corner's diagonal) and a common suffix of x bases (on the other corner's) with p + x >= the shorter
        // length. "prefix, one gap of |d|, suffix" then scores (shorter length) x match - (gap_open + (|d| - 1) gap_extend), which no
        // alignment of the two sides can beat; of the places the gap can take, the rows below leave it at the leftmost (their walk back
        // prefers the base step, which stays optimal as long as a place to its left is left) — opened once and extended, since opening
        // costs more than extending. Written straight away.
        const uint32_t lo = n < m ? n : m;
        uint32_t pre = lo, suf = lo;
        for (uint32_t c0 = 0; c0 < lo; c0 += 64) {
            const uint32_t x = c0 + lane;
            const unsigned long long bad = __ballot(x < lo && !S.eq(q0 + x, t0 + x));
            if (bad) { pre = c0 + static_cast<uint32_t>(__ffsll(static_cast<long long>(bad))) - 1u; break; }
        }
        for (uint32_t c0 = 0; c0 < lo; c0 += 64) {
            const uint32_t x = c0 + lane;
            const unsigned long long bad = __ballot(x < lo && !S.eq(q0 + n - 1 - x, t0 + m - 1 - x));
            if (bad) { suf = c0 + static_cast<uint32_t>(__ffsll(static_cast<long long>(bad))) - 1u; break; }
        }
        if (lo - suf <= pre) {
            const uint32_t at = lo - suf, gap = n < m ? m - n : n - m;
            C.put(7u, lo - at);
            C.put(n < m ? 2u : 1u, gap);
            C.put(7u, at);
            return SegOut{static_cast<int32_t>(lo) * V.match - (V.gap_open + static_cast<int32_t>(gap - 1) * V.gap_extend), 0u, m};
        }
    }
    for (uint32_t kk = lane; kk < W; kk += 64) { C.hrow[kk] = LNEG; C.frow[kk] = LNEG; }
    // the direction bytes of a short segment stay in LDS (most pieces: a few dozen rows), those of a long one go through the scratch
    const bool small = (static_cast<uint64_t>(n) + 1) * W <= C.tb_bytes;
    uint8_t* dstore = small ? C.tb : C.dirs;
    __syncthreads();
    int32_t bt = INT32_MIN; uint32_t bi = 0, bkk = 0;                     // this lane's best end node (free end)
    if (n_chunks == 1) {
        // a band of up to 64 diagonals (every extension, nearly every piece): the running rows stay in registers — the row above comes
        // over by a DPP shift — and a row needs no barrier
        const int32_t jb = static_cast<int32_t>(dlo) + static_cast<int32_t>(lane), mi = static_cast<int32_t>(m);
        const bool in = lane < W;
        const int32_t ext_at = V.gap_extend * static_cast<int32_t>(lane);
        int32_t h_reg = LNEG, f_reg = LNEG;
        uint32_t rb = 4u, ab = 5u;                                            // the bases of this row: read out of LDS a row ahead
        for (uint32_t i = 0; i <= n; i++) {
            if ((i & 63u) == 0) {
                __syncthreads();
                const uint32_t r = i + lane;
                C.rbuf[lane] = static_cast<uint8_t>(r >= 1 && r <= n ? S.read_base(q0 + r - 1) : 4u);
                const int64_t p0 = static_cast<int64_t>(t0) + static_cast<int64_t>(i) + dlo - 1;
                for (uint32_t x = lane; x < W + 64; x += 64) {
                    const int64_t p = p0 + x;
                    C.abuf[x] = static_cast<uint8_t>(p >= 0 && p < static_cast<int64_t>(S.alen) ? S.allele_base(static_cast<uint32_t>(p)) : 5u);
                }
                __syncthreads();
                rb = C.rbuf[0]; ab = C.abuf[lane];
            }
            const uint32_t rb_next = C.rbuf[(i + 1) & 63u], ab_next = C.abuf[((i + 1) & 63u) + lane];     // stale at a block's end: read again above
            const bool same = rb < 4u && rb == ab;
            const int32_t fr = free_start ? (i == 1 ? V.end_bonus : 0) : LNEG;
            const int32_t j = static_cast<int32_t>(i) + jb;
            const bool valid = in && j >= 0 && j <= mi;
            const int32_t up_h = from_right(h_reg, LNEG), up_f = from_right(f_reg, LNEG);
            int32_t mc = LNEG; uint32_t code = 0, mbit = 0;
            if (valid && i >= 1 && j >= 1) {
                int32_t base = h_reg;
                if (fr > base) { base = fr; code = 3; }
                if (base > LNEG / 2) {
                    mbit = same ? 1u : 0u;
                    mc = base + (same ? V.match : -V.mismatch);
                }
            }
            if (valid && i == 0 && j == 0 && !free_start) mc = 0;
            int32_t f = LNEG; uint32_t fbit = 0;
            if (valid && i >= 1) {
                const int32_t fo = up_h - V.gap_open, fe = up_f - V.gap_extend;
                if (fe > fo) { f = fe; fbit = 1; } else f = fo;
            }
            if (mc < LNEG / 2) mc = LNEG;
            if (f < LNEG / 2) f = LNEG;
            const int32_t ht = mc > f ? mc : f;
            const int32_t pm = prefix_max(valid && ht > LNEG ? ht + ext_at : INT32_MIN);
            const int32_t excl = from_left(pm, INT32_MIN);
            int32_t e = LNEG;
            if (valid && j >= 1 && excl != INT32_MIN) {
                e = excl - V.gap_open - (ext_at - V.gap_extend);
                if (e < LNEG / 2) e = LNEG;
            }
            int32_t h = mc;
            if (e > h) { h = e; code = 1; }
            if (f > h) { h = f; code = 2; }
            if (!valid) { h = LNEG; e = LNEG; f = LNEG; }
            const int32_t lh = from_left(h, LNEG), le = from_left(e, LNEG);
            const uint32_t ebit = valid && j >= 1 && le - V.gap_extend > lh - V.gap_open ? 1u : 0u;
            h_reg = h; f_reg = f;
            if (valid) dstore[static_cast<size_t>(i) * W + lane] = static_cast<uint8_t>(code | (ebit << 2) | (fbit << 3) | (mbit << 4));
            if (free_end && valid && mc > LNEG && i >= 1) {
                const int32_t total = mc + (i == n ? V.end_bonus : 0);
                if (total > bt) { bt = total; bi = i; bkk = lane; }
            }
            rb = rb_next; ab = ab_next;
        }
        if (in) C.hrow[lane] = h_reg;                                         // where the end node is looked up below
        __syncthreads();
    } else
    for (uint32_t i = 0; i <= n; i++) {
        if ((i & 63u) == 0) {
            // the bases of the next 64 rows: node (i, j) looks at read base q0 + i - 1 and allele base t0 + j - 1 (5: none)
            __syncthreads();
            const uint32_t r = i + lane;
            C.rbuf[lane] = static_cast<uint8_t>(r >= 1 && r <= n ? S.read_base(q0 + r - 1) : 4u);
            const int64_t p0 = static_cast<int64_t>(t0) + static_cast<int64_t>(i) + dlo - 1;
            for (uint32_t x = lane; x < W + 64; x += 64) {
                const int64_t p = p0 + x;
                C.abuf[x] = static_cast<uint8_t>(p >= 0 && p < static_cast<int64_t>(S.alen) ? S.allele_base(static_cast<uint32_t>(p)) : 5u);
            }
            __syncthreads();
        }
        const uint32_t rb = C.rbuf[i & 63u];
        const int32_t fr = free_start ? (i == 1 ? V.end_bonus : 0) : LNEG;
        int32_t carry_p = INT32_MIN, carry_h = LNEG, carry_e = LNEG;      // of the lanes to the left: best opener, H and deletion state of the last one
        for (uint32_t c = 0; c < n_chunks; c++) {
            const uint32_t kk = c * 64 + lane;
            const bool in = kk < W;
            const int64_t j = static_cast<int64_t>(i) + dlo + kk;
            const bool valid = in && j >= 0 && j <= static_cast<int64_t>(m);
            const int32_t old_h = in ? C.hrow[kk] : LNEG;
            const int32_t up_h = kk + 1 < W ? C.hrow[kk + 1] : LNEG, up_f = kk + 1 < W ? C.frow[kk + 1] : LNEG;
            int32_t mc = LNEG; uint32_t code = 0, mbit = 0;
            if (valid && i >= 1 && j >= 1) {
                int32_t base = old_h;
                if (fr > base) { base = fr; code = 3; }
                if (base > LNEG / 2) {
                    mbit = rb < 4u && rb == C.abuf[(i & 63u) + kk] ? 1u : 0u;
                    mc = base + (mbit ? V.match : -V.mismatch);
                }
            }
            if (valid && i == 0 && j == 0 && !free_start) mc = 0;
            int32_t f = LNEG; uint32_t fbit = 0;
            if (valid && i >= 1) {
                const int32_t fo = up_h - V.gap_open, fe = up_f - V.gap_extend;
                if (fe > fo) { f = fe; fbit = 1; } else f = fo;
            }
            if (mc < LNEG / 2) mc = LNEG;
            if (f < LNEG / 2) f = LNEG;
            // the deletions of the row: the best opener to the left, a prefix maximum over the lanes
            const int32_t ht = mc > f ? mc : f;
            int32_t pm = valid && ht > LNEG ? ht + V.gap_extend * static_cast<int32_t>(kk) : INT32_MIN;
            pm = prefix_max(pm);
            int32_t excl = from_left(pm, INT32_MIN);
            if (carry_p > excl) excl = carry_p;
            int32_t e = LNEG;
            if (valid && j >= 1 && excl != INT32_MIN) {
                e = excl - V.gap_open - V.gap_extend * (static_cast<int32_t>(kk) - 1);
                if (e < LNEG / 2) e = LNEG;
            }
            int32_t h = mc;
            if (e > h) { h = e; code = 1; }
            if (f > h) { h = f; code = 2; }
            if (!valid) { h = LNEG; e = LNEG; }
            const int32_t lh = from_left(h, carry_h), le = from_left(e, carry_e);
            const uint32_t ebit = valid && j >= 1 && le - V.gap_extend > lh - V.gap_open ? 1u : 0u;
            if (in) { C.hrow[kk] = h; C.frow[kk] = valid ? f : LNEG; }
            if (valid) dstore[static_cast<size_t>(i) * W + kk] = static_cast<uint8_t>(code | (ebit << 2) | (fbit << 3) | (mbit << 4));
            if (free_end && valid && mc > LNEG && i >= 1) {
                const int32_t total = mc + (i == n ? V.end_bonus : 0);
                if (total > bt) { bt = total; bi = i; bkk = kk; }
            }
            const int32_t last_p = __builtin_amdgcn_readlane(pm, 63);
            if (last_p > carry_p) carry_p = last_p;
            carry_h = __builtin_amdgcn_readlane(h, 63); carry_e = __builtin_amdgcn_readlane(e, 63);
        }
        __syncthreads();
    }
    uint32_t i, kk; int32_t score; uint32_t state;                      // 0: at H, 1: base step, 2: deletion, 3: insertion
    if (free_end) {
        // the first best end node by (i, diagonal) over the lanes; no extension at all unless one is better
        int32_t rt = bt; uint32_t ri = bi, rk = bkk;
        for (int o = 32; o > 0; o >>= 1) {
            const int32_t ot = __shfl_xor(rt, o); const uint32_t oi = static_cast<uint32_t>(__shfl_xor(static_cast<int>(ri), o)), ok = static_cast<uint32_t>(__shfl_xor(static_cast<int>(rk), o));
            if (ot > rt || (ot == rt && (oi < ri || (oi == ri && ok < rk)))) { rt = ot; ri = oi; rk = ok; }
        }
        const int32_t none = n == 0 ? V.end_bonus : 0;
        if (rt > none) { score = rt; i = ri; kk = rk; } else { score = none; i = 0; kk = static_cast<uint32_t>(-dlo); }
        state = 1;
    } else {
        i = n; kk = static_cast<uint32_t>(d - dlo);
        score = C.hrow[kk];
        if (free_start) {
            const int32_t fresh_all = n == 0 ? V.end_bonus : 0;
            if (!(score > fresh_all)) return SegOut{fresh_all, n, 0u};   // nothing before the first anchor is aligned
        }
        state = 0;
    }
    const uint32_t left = free_end ? n - i : 0u;
    const uint32_t j_end = static_cast<uint32_t>(static_cast<int64_t>(i) + dlo + kk);
    if (free_end) C.put(4u, left);                                         // the clip lies to the right of the extension
    const uint32_t rows_blk = C.tb_bytes / W;
    uint32_t blk_lo = small ? 0u : i + 1;                                  // rows [blk_lo, ..] of the direction bytes are in LDS (all of them: small)
    for (;;) {
        if (i < blk_lo) {
            __syncthreads();
            blk_lo = i + 1 > rows_blk ? i + 1 - rows_blk : 0u;
            const uint32_t nbytes = (i - blk_lo + 1) * W;
            const uint8_t* src = C.dirs + static_cast<size_t>(blk_lo) * W;
            for (uint32_t b = lane; b < nbytes; b += 64) C.tb[b] = src[b];
            __syncthreads();
        }
        const int64_t j = static_cast<int64_t>(i) + dlo + kk;
        const uint32_t dd = C.tb[(i - blk_lo) * W + kk];
        if (state == 0 && (dd & 3u) == 0 && i >= 1) {
            // a stretch of base steps that go on from H with the same outcome (= or X): the lanes look up the diagonal, the run is
            // taken at once (rows of this block, above row 0: what the walk below would do one node at a time)
            const uint32_t row = i - min(lane, i);
            const bool ok = lane <= i && row >= 1 && row >= blk_lo;
            const uint32_t b = ok ? C.tb[(row - blk_lo) * W + kk] : 0xFFu;
            const unsigned long long stop = __ballot(!(ok && (b & 0x13u) == (dd & 0x13u)));
            const uint32_t run = stop ? static_cast<uint32_t>(__ffsll(static_cast<long long>(stop))) - 1u : 64u;
            C.put((dd >> 4) & 1u ? 7u : 8u, run);
            i -= run;
            continue;
        }
        if (state == 0) {
            if (!free_start && i == 0 && j == 0) break;
            const uint32_t c = dd & 3u;
            state = (c == 0 || c == 3) ? 1u : (c == 1 ? 2u : 3u);
        }
        if (state == 1) {
            if (!free_start && i == 0 && j == 0) break;
            C.put((dd >> 4) & 1u ? 7u : 8u, 1u);
            i--;
            if ((dd & 3u) == 3u) return SegOut{score, i, m - static_cast<uint32_t>(static_cast<int64_t>(i) + dlo + kk)};      // started afresh
            state = 0;
        } else if (state == 2) {
            C.put(2u, 1u);
            kk--;
            state = (dd >> 2) & 1u ? 2u : 0u;
        } else {
            C.put(1u, 1u);
            i--; kk++;
            state = (dd >> 3) & 1u ? 3u : 0u;
        }
    }
    if (free_end) return SegOut{score, left, j_end};
    return SegOut{score, 0u, m};
}

__global__ __launch_bounds__(64) void map_long_align_kernel(const LongView V) {
    extern __shared__ int32_t lds_rows[];                                   // H row, F row [wmax each], then the bytes: direction block, allele bases, read bases
    const uint32_t lane = threadIdx.x;
    const uint32_t k = V.k;
    WaveState C{};
    C.hrow = lds_rows; C.frow = lds_rows + V.wmax; C.tb = reinterpret_cast<uint8_t*>(lds_rows + 2 * V.wmax); C.tb_bytes = V.tb_bytes;
    C.abuf = C.tb + V.tb_bytes; C.rbuf = C.abuf + V.wmax + 64;
    C.dirs = V.dirs + static_cast<size_t>(blockIdx.x) * V.dirs_bytes;
    C.ops = V.opsbuf + static_cast<size_t>(blockIdx.x) * V.ops_wave;
    for (;;) {
        // the next work item, whichever wavefront is free: the alignments of a chunk differ in length
        uint32_t w = 0;
        if (lane == 0) w = atomicAdd(&V.counters[4], 1u);
        w = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(w)));
        if (w >= V.n_work) break;
        const LongWork item = V.work[w];
        const uint64_t m = item.mate;
        const uint32_t g = item.g_slot & 0xFFFFu, slot = item.g_slot >> 16;
        const uint32_t strand = g & 1u, allele = V.basis[g >> 1];
        const uint32_t L = V.mate_len[m];
        const Seqs S{V.bases2, V.nmask, V.mate_off[m], V.seqs + V.seq_off[allele], V.allele_len[allele], strand};
        const uint2* chain = V.chain + item.chain_at;
        C.n = 0; C.cur_op = 0xFu; C.cur_len = 0;
        uint2 a = chain[0];
        SegOut r = wave_segment(V, S, a.x + k, L - (a.x + k), a.y + k, S.alen - (a.y + k), false, true, C);
        const uint32_t t_end = a.y + k + r.t_taken;
        int64_t score = static_cast<int64_t>(r.score) + static_cast<int64_t>(k) * V.match;
        C.put(7u, k);
        uint32_t cur_q = a.x, cur_t = a.y;
        for (uint32_t idx = 1; idx < item.chain_n; idx++) {
            a = chain[idx];
            if (a.x + k > cur_q || a.y + k > cur_t) {                           // overlapping seeds of one diagonal
                C.put(7u, cur_q - a.x);
                score += static_cast<int64_t>(cur_q - a.x) * V.match;
            } else {
                r = wave_segment(V, S, a.x + k, cur_q - (a.x + k), a.y + k, cur_t - (a.y + k), false, false, C);
                score += static_cast<int64_t>(r.score) + static_cast<int64_t>(k) * V.match;
                C.put(7u, k);
            }
            cur_q = a.x; cur_t = a.y;
        }
        r = wave_segment(V, S, 0u, cur_q, 0u, cur_t, true, false, C);
        score += r.score;
        C.put(4u, r.left);
        C.flush();
        const uint32_t t_start = cur_t - r.t_taken;
        // into the arena: the runs were met right to left in (q, t), which is left to right in BAM orientation on the reverse strand
        const uint32_t nw = C.n;
        uint32_t dst = 0;
        if (lane == 0) dst = atomicAdd(&V.counters[1], nw);
        dst = static_cast<uint32_t>(__shfl(static_cast<int>(dst), 0));
        __syncthreads();
        if (dst + nw <= V.ops_cap && dst + nw >= dst)
            for (uint32_t x = lane; x < nw; x += 64) V.ops[dst + x] = C.ops[strand ? x : nw - 1 - x];
        if (lane == 0) V.cands[m * V.slots + slot] = LongCand{static_cast<int32_t>(score), strand ? S.alen - t_end : t_start, dst, nw, g, 0u};
        __syncthreads();
    }
}

// ---- kernel 2: the records of a read end from its candidates; sizes (WRITE = false), then the records themselves
template <bool WRITE>
__device__ void map_long_emit_one(const LongView& V, const uint64_t m) {
    const uint32_t lane = threadIdx.x;
    const uint32_t L = V.mate_len[m];
    if (L == 0) {
        if (!WRITE && lane == 0) { V.n_recs[m] = 0; V.n_cigar[m] = 0; }
        return;
    }
    const uint64_t off = V.mate_off[m];
    const uint32_t nh = V.n_have[m];
    const LongCand* cands = V.cands + m * V.slots;
    // the primary record: best score, the smallest (allele, strand) on ties (the slots are in that order)
    int32_t top = INT32_MIN;
    for (uint32_t s = lane; s < nh; s += 64) top = max(top, cands[s].score);
    for (int o = 32; o > 0; o >>= 1) top = max(top, __shfl_xor(top, o));
    uint32_t prim = 0xFFFFFFFFu;
    for (uint32_t s = lane; s < nh; s += 64) if (cands[s].score == top) { prim = s; break; }
    for (int o = 32; o > 0; o >>= 1) prim = min(prim, static_cast<uint32_t>(__shfl_xor(static_cast<int>(prim), o)));
    const uint32_t ops_primary = nh ? cands[prim].n_words : 0u;
    const uint32_t mate2 = V.paired && (m & 1u) ? LCTY_FLAG_MATE2 : 0u;
    uint64_t rec0 = 0, cig0 = 0, rel0 = 0;
    if (WRITE) { rec0 = V.rec_at[m]; cig0 = V.cig_at[m]; rel0 = cig0 - V.pair_cig[m >> 1]; }
    uint32_t n_kept = 0, words = ops_primary, widest = 0;
    for (uint32_t s0 = 0; s0 < nh; s0 += 64) {
        const uint32_t s = s0 + lane;
        LongCand c{};
        if (s < nh) c = cands[s];
        const bool keep = s < nh && (s == prim || c.score >= V.min_score);
        const bool other = keep && s != prim;
        const unsigned long long om = __ballot(other);
        uint32_t incl = other ? c.n_words : 0u;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(incl, o); if (lane >= static_cast<uint32_t>(o)) incl += up; }
        if (keep) widest = max(widest, c.n_words);
        if (WRITE) {
            const uint32_t rank = s == prim ? 0u : 1u + n_kept + static_cast<uint32_t>(__popcll(om & ((1ull << lane) - 1ull)));
            const uint32_t cig_rel = s == prim ? 0u : words + incl - c.n_words;
            if (keep) {
                const uint32_t strand = c.g & 1u;
                const uint16_t flags = static_cast<uint16_t>((strand ? LCTY_FLAG_REVERSE : 0u) | (s == prim ? 0u : LCTY_FLAG_SECONDARY) | mate2);
                V.recs[rec0 + rank] = lcty_aln_rec{c.pos, V.basis[c.g >> 1], flags, c.n_words, static_cast<uint32_t>(rel0 + cig_rel)};
            }
            // the CIGAR words of the kept candidates of this batch, one candidate at a time over all lanes
            unsigned long long km = __ballot(keep);
            while (km) {
                const int src_lane = __ffsll(static_cast<long long>(km)) - 1;
                km &= km - 1;
                const uint32_t from = static_cast<uint32_t>(__shfl(static_cast<int>(c.ops_at), src_lane));
                const uint32_t nw = static_cast<uint32_t>(__shfl(static_cast<int>(c.n_words), src_lane));
                const uint32_t to = static_cast<uint32_t>(__shfl(static_cast<int>(cig_rel), src_lane));
                for (uint32_t w = lane; w < nw; w += 64) V.cigar[cig0 + to + w] = V.ops[from + w];
            }
        }
        n_kept += static_cast<uint32_t>(__popcll(om));
        words += static_cast<uint32_t>(__shfl(static_cast<int>(incl), 63));
    }
    if (!WRITE) {
        for (int o = 32; o > 0; o >>= 1) widest = max(widest, static_cast<uint32_t>(__shfl_xor(static_cast<int>(widest), o)));
        if (lane == 0) {
            V.n_recs[m] = nh ? n_kept + 1 : 1u;                                  // no candidate: one unmapped record
            V.n_cigar[m] = nh ? words : 0u;
            if (widest > V.counters[2]) atomicMax(&V.counters[2], widest);
        }
        return;
    }
    if (nh == 0 && lane == 0) V.recs[rec0] = lcty_aln_rec{0u, 0u, static_cast<uint16_t>(LCTY_FLAG_UNMAPPED | mate2), 0u, static_cast<uint32_t>(rel0)};
    // SEQ as the BAM has it: reverse-complemented when the primary record is on the reverse strand
    const bool primary_reverse = nh && (cands[prim].g & 1u);
    const uint32_t words16 = (L + 15) / 16;
    for (uint32_t wi = lane; wi < words16; wi += 64) {
        uint32_t out = 0;
        for (uint32_t j = 0; j < 16 && wi * 16 + j < L; j++) {
            const uint32_t i = wi * 16 + j, src = primary_reverse ? L - 1 - i : i;
            const uint32_t e = base_at(V.bases2, off, src);
            out |= (primary_reverse ? 3u - e : e) << (2 * j);
        }
        V.out_bases2[(off >> 4) + wi] = out;
    }
    for (uint32_t wi = lane; wi < (L + 31) / 32; wi += 64) {
        uint32_t out = 0;
        for (uint32_t j = 0; j < 32 && wi * 32 + j < L; j++) {
            const uint32_t i = wi * 32 + j, src = primary_reverse ? L - 1 - i : i;
            out |= static_cast<uint32_t>(n_at(V.nmask, off, src)) << j;
        }
        V.out_nmask[(off >> 5) + wi] = out;
    }
}

template <bool WRITE>
__global__ __launch_bounds__(64) void map_long_emit_kernel(const LongView V) {
    for (uint64_t m = blockIdx.x; m < V.n_mates; m += gridDim.x) map_long_emit_one<WRITE>(V, m);
}

}  // namespace

void run_map_long(lcty_locus* locus, const lcty_reads_host* chunk, const lcty_map_params* params, const MapIndex& ix, uint32_t max_len,
                  uint64_t* aln_off, uint64_t* cigar_off, bool sizes_only, MapRun& X) {
    lcty_ctx* ctx = locus->ctx;
    hipStream_t s = ctx->stream;
    const uint64_t n = chunk->n_pairs, n_mates = 2 * n;
    const uint64_t nb = chunk->mate_off[n_mates];
    // lcty_ctx_set_knob "map_trace" 1: wall-clock marks of the phases of a call on stderr
    const bool trace = ctx->diag_knob("map_trace", 0) != 0;
    const auto t_begin = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (trace) fprintf(stderr, "[lcty map] %8.3f ms %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(), what);
    };
    if (max_len > MAP_LONG_MAX_LEN) fail(LCTY_ERR_UNSUPPORTED, "read ends of up to %u bases (the longest here: %u)", MAP_LONG_MAX_LEN, max_len);
    if (params->chain_back == 0 || params->chain_back > 64) fail(LCTY_ERR_INVALID_INPUT, "chain_back %u: 1..64", params->chain_back);
    if (params->chain_gap == 0 || params->chain_gap > 8192) fail(LCTY_ERR_INVALID_INPUT, "chain_gap %u: 1..8192", params->chain_gap);
    if (params->chain_skew > 1024) fail(LCTY_ERR_INVALID_INPUT, "chain_skew %u: 0..1024", params->chain_skew);
    if (params->chain_skew > params->chain_gap) fail(LCTY_ERR_INVALID_INPUT, "chain_skew %u > chain_gap %u", params->chain_skew, params->chain_gap);
    LongView V{};
    V.table = ix.table.p; V.mask = ix.mask; V.entries = ix.entries.p; V.basis = ix.basis.p; V.n_basis = ix.n_basis;
    V.k = params->k; V.stride = params->stride; V.min_votes = std::max<uint32_t>(params->min_votes, 1);
    V.max_occ = params->max_occ ? params->max_occ : 4 * ix.n_basis;
    V.band = params->band; V.chain_gap = params->chain_gap; V.chain_skew = params->chain_skew; V.chain_back = params->chain_back;
    V.match = params->match; V.mismatch = params->mismatch; V.end_bonus = params->end_bonus; V.min_score = params->min_score;
    V.gap_open = params->gap_open; V.gap_extend = params->gap_extend;
    V.seqs = locus->d_seqs.p; V.seq_off = locus->d_seq_off.p; V.allele_len = locus->d_allele_len.p;
    V.n_mates = n_mates; V.mate_len = X.d_len.p; V.mate_off = X.d_off.p; V.bases2 = X.d_b2.p; V.nmask = X.d_nm.p;
    V.paired = locus->bg.is_paired;
    V.n_recs = X.d_nrec.p; V.n_cigar = X.d_ncig.p;
    if (params->gap_open < params->gap_extend) fail(LCTY_ERR_UNSUPPORTED, "the long route needs gap_open >= gap_extend (%d < %d)", params->gap_open, params->gap_extend);
    const uint32_t n_groups = 2 * ix.n_basis;
    const uint32_t cus = static_cast<uint32_t>(ctx->props.multiProcessorCount);
    size_t free_b = 0, total_b = 0;
    V.slots = n_groups;
    if (n_mates * V.slots > 0xFFFFFFFFull) fail(LCTY_ERR_UNSUPPORTED, "chunks of up to %llu read pairs with this basis", (unsigned long long)(0xFFFFFFFFull / V.slots / 2));
    X.d_cands.ensure_slack(n_mates * V.slots * sizeof(LongCand)); X.d_nhave.ensure_slack(n_mates);
    X.d_counters.ensure_slack(8);
    V.cands = reinterpret_cast<LongCand*>(X.d_cands.p); V.n_have = X.d_nhave.p; V.counters = X.d_counters.p;
    uint32_t counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // ---- kernel 1: the chains. Scratch of a workgroup: the anchors of every group; it leaves the chains to be aligned (repeated with
    // more room if they did not fit) and a work item for each
    const uint32_t max_seeds = max_len >= params->k ? (max_len - params->k) / params->stride + 2 : 0;
    V.cap_g = std::max<uint32_t>(2 * max_seeds, 1);
    {
        const uint64_t per_wg = static_cast<uint64_t>(n_groups) * V.cap_g * sizeof(uint4);
        LCTY_HIP(hipMemGetInfo(&free_b, &total_b));
        if (per_wg > free_b / 2) fail(LCTY_ERR_RUNTIME, "the long route needs %llu MB of scratch per workgroup", (unsigned long long)(per_wg >> 20));
        const uint64_t budget = std::min<uint64_t>(free_b / 4, 16ull << 30);
        const uint32_t n_wg = static_cast<uint32_t>(std::min<uint64_t>(std::min<uint64_t>(n_mates, 8ull * cus), std::max<uint64_t>(budget / per_wg, 1)));
        X.d_anchors.ensure_slack(static_cast<size_t>(n_wg) * n_groups * V.cap_g);
        V.anchors = X.d_anchors.p;
        X.d_work.ensure_slack(n_mates * V.slots * (sizeof(LongWork) / sizeof(uint32_t)));
        V.work = reinterpret_cast<LongWork*>(X.d_work.p);
        uint64_t seeds_total = 0;
        for (uint64_t m = 0; m < n_mates; m++)
            if (chunk->mate_len[m] >= params->k) seeds_total += (chunk->mate_len[m] - params->k) / params->stride + 2;
        // room for the chains of a chunk of noisy reads on every basis allele (every other seed an anchor), as far as the memory goes; the
        // kernel says how many entries it needed when that was not enough
        LCTY_HIP(hipMemGetInfo(&free_b, &total_b));
        uint64_t cap = std::min<uint64_t>(0xFFFFFFF0ull, std::min<uint64_t>(free_b / 64, seeds_total * ix.n_basis / 2) + 4096);
        for (;;) {
            X.d_counters.zero(s);
            X.d_chain.ensure_slack(cap);
            V.chain = X.d_chain.p; V.chain_cap = static_cast<uint32_t>(cap);
            ctx->timed(LCTY_K_MAP, [&] { hipLaunchKernelGGL(map_long_chain_kernel, dim3(n_wg), dim3(64), 0, s, V); }, s);
            LCTY_HIP(hipGetLastError());
            X.d_counters.download(counters, 4, s);
            LCTY_HIP(hipStreamSynchronize(s));
            if (counters[3] <= cap) break;
            if (counters[3] >= 0xFFFFFFF0u) fail(LCTY_ERR_UNSUPPORTED, "anchors of the chunk's chains: map it in parts");
            cap = static_cast<uint64_t>(counters[3]) + 1024;
        }
        mark("chains done");
        V.n_work = counters[0];
    }
    // ---- kernel 2: the alignments. Scratch of a wavefront: the direction bytes of the largest segment, the CIGAR runs of a candidate;
    // it leaves the CIGAR words in an arena (repeated with more room if they did not fit)
    if (V.n_work) {
        V.wmax = params->chain_skew + 2 * params->band + 1;
        V.tb_bytes = std::max<uint32_t>(4096, (V.wmax + 3) / 4 * 4);
        const uint64_t ext_bytes = (static_cast<uint64_t>(max_len) + 1) * (2 * params->band + 1);
        const uint64_t piece_rows = std::min<uint64_t>(params->chain_gap, max_len) + 1;
        V.dirs_bytes = (std::max<uint64_t>(ext_bytes, piece_rows * V.wmax) + 15) / 16 * 16;
        V.ops_wave = 2 * max_len + 8;
        const uint64_t per_wave = V.dirs_bytes + static_cast<uint64_t>(V.ops_wave) * 4;
        LCTY_HIP(hipMemGetInfo(&free_b, &total_b));
        if (per_wave > free_b / 2) fail(LCTY_ERR_RUNTIME, "the long route needs %llu MB of scratch per wavefront", (unsigned long long)(per_wave >> 20));
        const uint64_t budget = std::min<uint64_t>(free_b / 4, 16ull << 30);
        const uint32_t lds = 2 * V.wmax * 4 + V.tb_bytes + (V.wmax + 64) + 64;
        const uint32_t n_waves = static_cast<uint32_t>(std::min<uint64_t>(std::min<uint64_t>(V.n_work, 24ull * cus), std::max<uint64_t>(budget / per_wave, 1)));
        X.d_dirs.ensure_slack(static_cast<size_t>(n_waves) * V.dirs_bytes);
        X.d_opsbuf.ensure_slack(static_cast<size_t>(n_waves) * V.ops_wave);
        V.dirs = X.d_dirs.p; V.opsbuf = X.d_opsbuf.p;
        // room for the words of a chunk of noisy reads (a word per ~4 bases of every alignment), as far as the memory goes; the kernel
        // says how many it needed when that was not enough
        LCTY_HIP(hipMemGetInfo(&free_b, &total_b));
        uint64_t cap = std::min<uint64_t>(0xFFFFFFF0ull, std::min<uint64_t>(free_b / 32, static_cast<uint64_t>(V.n_work) * (nb / std::max<uint64_t>(n, 1) / 4 + 8)) + 4096);
        mark("alignment scratch allocated");
        LCTY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(map_long_align_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        for (;;) {
            const uint32_t zero = 0;
            X.d_counters.upload(&zero, 1, s, 1); X.d_counters.upload(&zero, 1, s, 4);
            X.d_ops.ensure_slack(cap);
            V.ops = X.d_ops.p; V.ops_cap = static_cast<uint32_t>(cap);
            ctx->timed(LCTY_K_MAP, [&] { hipLaunchKernelGGL(map_long_align_kernel, dim3(n_waves), dim3(64), lds, s, V); }, s);
            LCTY_HIP(hipGetLastError());
            X.d_counters.download(counters, 4, s);
            LCTY_HIP(hipStreamSynchronize(s));
            if (counters[1] <= cap) break;
            if (counters[1] >= 0xFFFFFFF0u) fail(LCTY_ERR_UNSUPPORTED, "CIGAR words of the chunk's alignments: map it in parts");
            cap = static_cast<uint64_t>(counters[1]) + 1024;
        }
        mark("alignments done");
    } else {
        X.d_ops.ensure_slack(1);
        V.ops = X.d_ops.p; V.ops_cap = 1;
    }
    // ---- kernel 3, sizes
    const uint32_t n_wg2 = static_cast<uint32_t>(std::min<uint64_t>(n_mates, 16ull * cus));
    ctx->timed(LCTY_K_MAP, [&] { hipLaunchKernelGGL(map_long_emit_kernel<false>, dim3(n_wg2), dim3(64), 0, s, V); }, s);
    LCTY_HIP(hipGetLastError());
    X.d_counters.download(counters, 4, s);
    X.nrec.resize(n_mates); X.ncig.resize(n_mates);
    X.d_nrec.download(X.nrec.data(), n_mates, s); X.d_ncig.download(X.ncig.data(), n_mates, s);
    LCTY_HIP(hipStreamSynchronize(s));
    X.max_rec_cigar = counters[2];
    mark("sizes done");
    std::vector<uint64_t> rec_at(n_mates), cig_at(n_mates), pair_cig(n);
    uint64_t r = 0, c = 0;
    for (uint64_t p = 0; p < n; p++) {
        pair_cig[p] = c;
        for (uint32_t e = 0; e < 2; e++) { rec_at[2 * p + e] = r; cig_at[2 * p + e] = c; r += X.nrec[2 * p + e]; c += X.ncig[2 * p + e]; }
        aln_off[p + 1] = r; cigar_off[p + 1] = c;
    }
    X.n_recs = r; X.n_cigar = c;
    if (sizes_only) return;
    X.d_rec_at.ensure_slack(n_mates); X.d_rec_at.upload(rec_at.data(), n_mates, s);
    X.d_cig_at.ensure_slack(n_mates); X.d_cig_at.upload(cig_at.data(), n_mates, s);
    X.d_pair_cig.ensure_slack(n); X.d_pair_cig.upload(pair_cig.data(), n, s);
    X.d_recs.ensure_slack(std::max<uint64_t>(r, 1)); X.d_cigar.ensure_slack(std::max<uint64_t>(c, 1));
    X.d_ob2.ensure_slack(std::max<uint64_t>(nb / 16, 1)); X.d_onm.ensure_slack(std::max<uint64_t>(nb / 32, 1));
    X.d_ob2.zero(s); X.d_onm.zero(s);
    V.rec_at = X.d_rec_at.p; V.cig_at = X.d_cig_at.p; V.pair_cig = X.d_pair_cig.p; V.recs = X.d_recs.p; V.cigar = X.d_cigar.p;
    V.out_bases2 = X.d_ob2.p; V.out_nmask = X.d_onm.p;
    ctx->timed(LCTY_K_MAP, [&] { hipLaunchKernelGGL(map_long_emit_kernel<true>, dim3(n_wg2), dim3(64), 0, s, V); }, s);
    LCTY_HIP(hipGetLastError());
    LCTY_HIP(hipStreamSynchronize(s));                                          // rec_at & co. are host vectors of this frame
    mark("records written");
}

}  // namespace lcty
