// lcty_score.hip — AllAlignments::load (src/model/locs.rs:1085-1185, 1237-1288, no alignment
// recovery) as ONE fused gfx950 kernel: one 64-lane wavefront per read pair.
//
// Each phase is organised as wide, independent memory operations (a naive lane-chases-its-records
// version is bound by serial global-memory latency):
//
//   prologue   both mates' 2-bit words are prefetched into registers (one 64-bit word per lane);
//              lanes 0/1 score the two primaries and derive the edit thresholds
//              (EditDistCache + poor-complexity relaxation, locs.rs:529-536). The index of the
//              mate-2 primary comes from the host (pair_meta), so thresholds are known first.
//   pass 1     lanes over BAM records in groups of 4x64: 4 record loads, then 4x2 unaligned
//              dwordx4 CIGAR loads in flight, branch-light op counting (aln.rs:301-317),
//              ErrorProfile::ln_prob (err_prof.rs:212-221), per-end best edit / ln-prob in
//              registers. Saved alignments (locs.rs:310-314) go to LDS (16 B each) and are
//              chained per (contig, read end) with one LDS exchange (no sort, no scan).
//   K2         unique k-mers (locs.rs:968-1002): windows extracted with cross-lane reads from the
//              prefetched words, hash probes issued 4 at a time, ballot + greedy ctz walk.
//   pass 3a    lane per contig. 1x1 groups (the common case) are paired entirely in registers;
//              anything else takes the general path: (ln_prob desc) order, 128-bp dedupe
//              (locs.rs:321-342), top-10, full pair enumeration (746-799). Writes the matrix row
//              (locs.rs:1203-1212) and the per-contig kept count; in_bounds (1008-1014).
//   pass 3b    PairAlignments emitted contig-ascending into the arena (one atomicAdd per pair).
//
// Launch: 64-thread workgroups (one wave), grid-strided over pairs; dynamic LDS sized from the
// largest record count of any pair in the batch. All LDS traffic is wave-private.
// (Staging a pair's records + CIGAR words through LDS with all loads in flight was measured: slower —
//  pass 1 is instruction-issue bound, not latency bound, once the loads are grouped.)
#include "lcty_objects.hpp"

namespace lcty {

constexpr uint32_t NONE32 = 0xFFFFFFFFu;
constexpr uint32_t NONE16 = 0xFFFFu;
constexpr uint32_t REV_BIT = 0x80000000u;
constexpr int GR = 4;                         // 64-record chunks whose loads are kept in flight together

// saved alignment in LDS: ln_prob (not yet normalised), start, end | strand<<31
struct __attribute__((aligned(16))) Rec16 {
    double ln_prob;
    uint32_t start, end_rev;
};
static_assert(sizeof(Rec16) == 16, "Rec16 layout");

struct __attribute__((packed, aligned(4))) W8 { uint32_t w[8]; };
struct __attribute__((packed, aligned(4))) W4 { uint32_t w[4]; };

__device__ inline uint32_t wave_min_u32(uint32_t v) {
    for (int o = WAVE / 2; o > 0; o >>= 1) v = min(v, static_cast<uint32_t>(__shfl_xor(static_cast<int>(v), o)));
    return v;
}
__device__ inline double wave_max_f64(double v) {
    for (int o = WAVE / 2; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ inline uint32_t wave_sum_u32(uint32_t v) {
    for (int o = WAVE / 2; o > 0; o >>= 1) v += static_cast<uint32_t>(__shfl_xor(static_cast<int>(v), o));
    return v;
}
__device__ inline double wave_sum_f64(double v) {
    for (int o = WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// exclusive prefix sum across the wave; *total = wave sum
__device__ inline uint32_t wave_excl_scan_u32(uint32_t v, int lane, uint32_t* total) {
    uint32_t x = v;
    for (int o = 1; o < WAVE; o <<= 1) {
        const uint32_t y = static_cast<uint32_t>(__shfl_up(static_cast<int>(x), o));
        if (lane >= o) x += y;
    }
    *total = static_cast<uint32_t>(__shfl(static_cast<int>(x), WAVE - 1));
    return x - v;
}

// cold path: insert sizes beyond the cached range are evaluated directly (nbinom.rs:128-131)
__device__ __noinline__ double nbinom_ln_pmf_direct(double n, double lnq, double lnpmf_const, uint32_t sz) {
    const double x = static_cast<double>(sz);
    return lnpmf_const + lgamma(n + x) - lgamma(x + 1.0) + x * lnq;
}
// InsertDistr::ln_prob -> LinearCache::ln_pmf (insertsz.rs:153-155, lincache.rs:41-48).
// Held by value: the kernel-argument structs must never be addressed (that would spill them to scratch).
struct InsLut {
    const double* lut;
    uint32_t size;
    double n, lnq, lnpmf_const;
    __device__ __forceinline__ double ln_prob(uint32_t sz) const {
        if (sz < size) return lut[sz];
        return nbinom_ln_pmf_direct(n, lnq, lnpmf_const, sz);
    }
};

// ---------------------------------------------------------------------------------------------
// CIGAR -> operation counts (count_region_operations_fast, aln.rs:301-317; soft_clipping,
// cigar.rs:519-527; hard_to_soft, cigar.rs:309-320).
// ---------------------------------------------------------------------------------------------
struct OpCounts {
    uint32_t matches, mism, ins, del, left, right;
    bool bad;
};

// One CIGAR word. Clipping is handled by the caller from the first / last word, so here S and H only
// have to be accepted (H: first / last word only; anywhere else it panics like any unsupported op, aln.rs:311).
__device__ __forceinline__ void count_op(OpCounts& c, uint32_t w, bool edge) {
    const uint32_t op = w & 15u;
    const uint32_t len = w >> 4;
    c.matches += op == LCTY_CIGAR_EQ ? len : 0u;
    c.mism += op == LCTY_CIGAR_X ? len : 0u;
    c.del += op == LCTY_CIGAR_D ? len : 0u;
    c.ins += op == LCTY_CIGAR_I ? len : 0u;
    const uint32_t valid = edge ? 0x1B6u : 0x196u;      // I, D, S, =, X (+ H at the ends, turned into S)
    c.bad |= ((valid >> op) & 1u) == 0u;
}

// first 8 words are already in registers; longer CIGARs continue from memory
__device__ __forceinline__ OpCounts count_ops(const W8& first, const uint32_t* cig, uint32_t nc, bool primary) {
    OpCounts c{0, 0, 0, 0, 0, 0, false};
    uint32_t wl = first.w[0];
#pragma unroll
    for (uint32_t i = 0; i < 8; i++)
        if (i < nc) { count_op(c, first.w[i], i == 0 || i + 1 == nc); wl = first.w[i]; }
    // long CIGARs (ONT / HiFi): eight, then four words per step from unaligned 16-byte loads
    uint32_t i = 8;
    for (; i + 8 <= nc; i += 8) {
        const W8 q = *reinterpret_cast<const W8*>(cig + i);
#pragma unroll
        for (uint32_t t = 0; t < 8; t++) count_op(c, q.w[t], i + t + 1 == nc);
        wl = q.w[7];
    }
    for (; i + 4 <= nc; i += 4) {
        const W4 q = *reinterpret_cast<const W4*>(cig + i);
#pragma unroll
        for (uint32_t t = 0; t < 4; t++) count_op(c, q.w[t], i + t + 1 == nc);
        wl = q.w[3];
    }
    for (; i < nc; i++) { wl = cig[i]; count_op(c, wl, i + 1 == nc); }
    // soft_clipping (cigar.rs:519-527) after hard_to_soft (309-320)
    const uint32_t w0 = first.w[0];
    const uint32_t op0 = w0 & 15u, opl = wl & 15u;
    if (op0 == LCTY_CIGAR_S || op0 == LCTY_CIGAR_H) c.left = w0 >> 4;
    if (opl == LCTY_CIGAR_S || opl == LCTY_CIGAR_H) c.right = wl >> 4;
    if (primary && (op0 == LCTY_CIGAR_H || opl == LCTY_CIGAR_H)) c.bad = true;   // assert!(!cigar.has_hard_clipping()), locs.rs:526
    return c;
}

struct Scored {
    double ln_prob;
    uint32_t start, end, edit;
    bool bad;
};

// An alignment whose operations have been counted by the caller (lcty_aln_counted, 16 bytes): the OperCounts that
// count_region_operations_fast + limited_clipping leave (aln.rs:288-317), so only edit_distance (err_prof.rs:73-79) and
// ErrorProfile::ln_prob (212-221) remain. raw = {pos | flags << 28, contig | matches << 16, mismatches | insertions << 16,
// deletions | clipping << 16}
__device__ __forceinline__ Scored score_counted(const LocusView& L, const uint4 raw) {
    const uint32_t matches = raw.y >> 16, mism = raw.z & 0xFFFFu, ins = raw.z >> 16, del = raw.w & 0xFFFFu, clip = raw.w >> 16;
    Scored s;
    s.start = raw.x & 0x0FFFFFFFu;
    s.end = s.start + matches + mism + del;
    const uint32_t common = mism + ins + clip;
    s.edit = common + del;
    s.ln_prob = L.lp[0] * static_cast<double>(matches) + L.lp[1] * static_cast<double>(mism)
              + L.lp[2] * static_cast<double>(ins) + L.lp[3] * static_cast<double>(del)
              + L.lp[4] * static_cast<double>(clip);
    s.bad = false;
    return s;
}

__device__ __forceinline__ Scored score_counts(const LocusView& L, const OpCounts& c, uint32_t pos, uint32_t contig_len) {
    Scored s;
    const uint32_t ref_len = c.matches + c.mism + c.del;
    s.start = pos;
    s.end = pos + ref_len;
    const uint32_t clip = min(c.left, pos) + min(c.right, contig_len > s.end ? contig_len - s.end : 0u);   // aln.rs:288-296
    const uint32_t common = c.mism + c.ins + clip;                                                          // err_prof.rs:73-79
    s.edit = common + c.del;
    s.ln_prob = L.lp[0] * static_cast<double>(c.matches) + L.lp[1] * static_cast<double>(c.mism)
              + L.lp[2] * static_cast<double>(c.ins) + L.lp[3] * static_cast<double>(c.del)
              + L.lp[4] * static_cast<double>(clip);                                                        // err_prof.rs:212-221
    s.bad = c.bad;
    return s;
}

// ---------------------------------------------------------------------------------------------
// K2: unique k-mers (locs.rs:976-993).
// ---------------------------------------------------------------------------------------------
struct KmerWalk {
    uint32_t count, next_allowed;
    __device__ inline void feed(unsigned long long m, uint32_t base, uint32_t k) {
        // greedy: take a hit, then skip the next k-1 k-mers (`kmers_iter.nth(k_2)`, locs.rs:988)
        if (next_allowed > base) {
            const uint32_t sh = next_allowed - base;
            m = sh >= 64 ? 0ull : (m >> sh) << sh;
        }
        while (m) {
            const uint32_t b = static_cast<uint32_t>(__ffsll(static_cast<long long>(m))) - 1u;
            count = count == 0xFFFFu ? count : count + 1;         // saturating_add
            next_allowed = base + b + k;
            const uint32_t sh = b + k;
            m = sh >= 64 ? 0ull : (m >> sh) << sh;
        }
    }
};

// general path: windows straight from global memory (mates longer than 2016 bases)
__device__ __noinline__ uint32_t mate_unique_kmers_global(const uint64_t* kset, uint64_t kset_mask, uint32_t undef_in_set, uint32_t k,
                                                         const uint64_t* w64, const uint32_t* nm, uint32_t len, int lane) {
    if (len < k) return 0;
    const uint32_t nk = len + 1 - k;
    KmerWalk walk{0, 0};
    for (uint32_t base = 0; base < nk; base += WAVE) {
        const uint32_t q = base + lane;
        bool hit = false;
        if (q < nk) {
            if (k > 31) {                                                // 128-bit k-mers: a table of {lo, hi} pairs
                if (window_has_n_wide(nm, q, k)) hit = undef_in_set != 0;
                else hit = kset128_contains(kset, kset_mask, canonical_kmer_2bit128(w64, q, k));
            } else if (window_has_n(nm, q, k)) hit = undef_in_set != 0;  // UNDEF k-mer (kmers.rs:184-190)
            else hit = kset_contains(kset, kset_mask, canonical_kmer_2bit(w64, q, k));
        }
        walk.feed(__ballot(hit), base, k);
    }
    return walk.count;
}

// canonical k-mer (kmers.rs:192-196) of the window that starts `sh`/2 bases into the 32-base word `lo`
__device__ __forceinline__ uint64_t canon_from_words(uint64_t lo, uint64_t hi, uint32_t sh, uint32_t k) {
    uint64_t x = lo >> sh;
    if (sh) x |= hi << (64u - sh);
    const uint64_t mask = (1ull << (2u * k)) - 1ull;
    x &= mask;
    const uint64_t rv = (~x) & mask;
    uint64_t y = __brevll(x);
    y = ((y >> 1) & 0x5555555555555555ull) | ((y & 0x5555555555555555ull) << 1);
    const uint64_t fw = y >> (64u - 2u * k);
    return rv < fw ? rv : fw;
}

// Both mates at once from register-resident words (lane l holds 64-bit word l and N-mask word l of each
// mate). Hash probes of up to 4 (mate, 64-window chunk) items are in flight together.
__device__ __forceinline__ void pair_unique_kmers_regs(const uint64_t* kset, uint64_t kset_mask, uint32_t undef_in_set, uint32_t k,
                                                       uint32_t len0, uint32_t len1, uint64_t bw0, uint64_t bw1,
                                                       uint32_t nm0, uint32_t nm1, int lane, uint32_t* uk0, uint32_t* uk1) {
    const uint32_t nk0 = len0 >= k ? len0 + 1 - k : 0u, nk1 = len1 >= k ? len1 + 1 - k : 0u;
    const uint32_t nch0 = (nk0 + WAVE - 1) / WAVE, nch1 = (nk1 + WAVE - 1) / WAVE;
    const uint32_t items = nch0 + nch1;
    KmerWalk walk0{0, 0}, walk1{0, 0};
    const uint32_t half = static_cast<uint32_t>(lane) >> 5;
    const uint32_t sh2 = (lane & 31u) * 2u, sh1 = lane & 31u;
    for (uint32_t it0 = 0; it0 < items; it0 += 4) {
        uint64_t key[4], slot[4];
        bool live[4], undef[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t it = it0 + j;
            const bool m = it >= nch0;                    // wave-uniform
            const uint32_t t = m ? it - nch0 : it;
            const uint32_t nk = m ? nk1 : nk0;
            const uint32_t q = t * WAVE + lane;
            live[j] = it < items && q < nk;
            // 128-base span of this chunk = words 2t, 2t+1, 2t+2 (wave-uniform indices)
            const uint64_t src = m ? bw1 : bw0;
            const uint32_t nsrc = m ? nm1 : nm0;
            // the three words of the chunk come from lanes whose index is the same for the whole wavefront: v_readlane (a scalar
            // operand for what follows), not a trip through the LDS crossbar
            const int wi = __builtin_amdgcn_readfirstlane(static_cast<int>((2 * t) & 63u));
            auto lane64 = [](uint64_t v, int l) -> uint64_t {
                const uint32_t lo = __builtin_amdgcn_readlane(static_cast<uint32_t>(v), l), hi = __builtin_amdgcn_readlane(static_cast<uint32_t>(v >> 32), l);
                return (static_cast<uint64_t>(hi) << 32) | lo;
            };
            const uint64_t wA = lane64(src, wi), wB = lane64(src, (wi + 1) & 63), wC = lane64(src, (wi + 2) & 63);
            const uint32_t nA = __builtin_amdgcn_readlane(nsrc, wi), nB = __builtin_amdgcn_readlane(nsrc, (wi + 1) & 63), nC = __builtin_amdgcn_readlane(nsrc, (wi + 2) & 63);
            key[j] = canon_from_words(half ? wB : wA, half ? wC : wB, sh2, k);
            uint32_t nbits = (half ? nB : nA) >> sh1;
            if (sh1 + k > 32u) nbits |= (half ? nC : nB) << (32u - sh1);
            undef[j] = (nbits & ((1u << k) - 1u)) != 0u;  // window contains a non-ACGT base -> UNDEF (kmers.rs:184-190)
            slot[j] = mix64(key[j]) & kset_mask;
        }
        // The probes of all four items advance together, two slots of the table per step: a step costs one trip to the L2 whatever
        // the number of lanes and items still searching (one item after the other, one slot at a time, the four chains add up:
        // at a load of 1/4 the longest chain of 64 lanes is 4 or 5 slots).
        uint32_t pend = 0, hits = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (live[j] && !undef[j]) pend |= 1u << j;
            if (live[j] && undef[j] && undef_in_set != 0) hits |= 1u << j;
        }
        while (__any(pend != 0)) {
            uint64_t v0[4], v1[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool go = (pend >> j) & 1u;
                v0[j] = go ? kset[slot[j]] : KSET_EMPTY;
                v1[j] = go ? kset[(slot[j] + 1) & kset_mask] : KSET_EMPTY;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (!((pend >> j) & 1u)) continue;
                const bool found = v0[j] == key[j] || (v0[j] != KSET_EMPTY && v1[j] == key[j]);
                const bool ended = v0[j] == KSET_EMPTY || v1[j] == KSET_EMPTY;
                if (found) hits |= 1u << j;
                if (found || ended) pend &= ~(1u << j);
                else slot[j] = (slot[j] + 2) & kset_mask;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t it = it0 + j;
            if (it < items) {                             // wave-uniform
                const unsigned long long mask = __ballot((hits >> j) & 1u);
                const bool m = it >= nch0;
                const uint32_t t = m ? it - nch0 : it;
                if (m) walk1.feed(mask, t * WAVE, k); else walk0.feed(mask, t * WAVE, k);
            }
        }
    }
    *uk0 = walk0.count;
    *uk1 = walk1.count;
}

// ---------------------------------------------------------------------------------------------
// Pairing (identify_contig_pair_alns, locs.rs:746-799 / identify_single_end_alignments 873-911)
// ---------------------------------------------------------------------------------------------
struct AlnRef {
    double lp;
    uint32_t start, end, idx;
    bool rev;
};

__device__ __forceinline__ AlnRef load_aln(const Rec16* rec, uint32_t idx, double best_lik) {
    const Rec16 r = rec[idx];
    return AlnRef{r.ln_prob - best_lik, r.start, r.end_rev & ~REV_BIT, idx, (r.end_rev & REV_BIT) != 0};   // normalize_probs, locs.rs:358-360
}

template <typename C, typename F> __device__ inline void enumerate_pairs(const C& c, F&& f);

struct PairCtx {
    InsLut ins;
    const Rec16* rec;
    const uint16_t* ord1;
    const uint16_t* ord2;
    uint32_t k1, k2;
    double best0, best1;
    double unm_ins_penalty;
    bool paired;

    __device__ inline AlnRef get1(uint32_t i) const { return load_aln(rec, ord1[i], best0); }
    __device__ inline AlnRef get2(uint32_t j) const { return load_aln(rec, ord2[j], best1); }
    __device__ inline double pair_prob(const AlnRef& a1, const AlnRef& a2) const {
        // paired_prob (aln.rs:236-238) with furthest_distance (interv.rs:179-185)
        const uint32_t insert = max(a1.end, a2.end) - min(a1.start, a2.start);
        return a1.lp + a2.lp + ins.ln_prob(insert);
    }
    // f(prob, order, aln1 | idx NONE32, aln2 | idx NONE32) for every PairAlignment pushed, in push order
    template <typename F>
    __device__ inline void enumerate(F&& f) const { enumerate_pairs(*this, f); }
};

// identify_contig_pair_alns (locs.rs:746-799) over any context that has k1, k2, paired, unm_ins_penalty, get1, get2, pair_prob
template <typename C, typename F>
__device__ inline void enumerate_pairs(const C& c, F&& f) {
    const AlnRef none{0.0, 0, 0, NONE32, false};
    if (!c.paired) {
        for (uint32_t i = 0; i < c.k1; i++) { const AlnRef a = c.get1(i); f(a.lp, i, a, none); }
        return;
    }
    for (uint32_t i = 0; i < c.k1; i++) {
        const AlnRef a1 = c.get1(i);
        double m1 = -INFINITY;
        for (uint32_t j = 0; j < c.k2; j++) {
            const AlnRef a2 = c.get2(j);
            if (a1.rev != a2.rev) {
                const double prob = c.pair_prob(a1, a2);
                if (isfinite(prob)) { m1 = fmax(m1, prob); f(prob, i * (c.k2 + 1) + j, a1, a2); }
            }
        }
        const double alone = a1.lp + c.unm_ins_penalty;
        if (alone >= m1) f(alone, i * (c.k2 + 1) + c.k2, a1, none);
    }
    for (uint32_t j = 0; j < c.k2; j++) {
        const AlnRef a2 = c.get2(j);
        double m2 = -INFINITY;
        for (uint32_t i = 0; i < c.k1; i++) {
            const AlnRef a1 = c.get1(i);
            if (a1.rev != a2.rev) {
                const double prob = c.pair_prob(a1, a2);
                if (isfinite(prob)) m2 = fmax(m2, prob);
            }
        }
        const double alone = a2.lp + c.unm_ins_penalty;
        if (alone >= m2) f(alone, c.k1 * (c.k2 + 1) + j, none, a2);
    }
}

// the same pairing over at most two saved alignments per read end, held in registers (the lean kernel's second pass over the pairs it
// had left: score_lean_body<.., TWO>)
struct SmallCtx {
    InsLut ins;
    AlnRef a10, a11, a20, a21;          // named members, selects: an array whose address is taken lives in scratch memory
    uint32_t k1, k2;
    double unm_ins_penalty;
    bool paired;
    __device__ inline AlnRef get1(uint32_t i) const { return i ? a11 : a10; }
    __device__ inline AlnRef get2(uint32_t j) const { return j ? a21 : a20; }
    __device__ inline double pair_prob(const AlnRef& x1, const AlnRef& x2) const {
        const uint32_t insert = max(x1.end, x2.end) - min(x1.start, x2.start);
        return x1.lp + x2.lp + ins.ln_prob(insert);
    }
};

// General path, one (contig, end) group: copy the chain into `ord`, order by (ln_prob desc, record
// index asc), drop later members of a 128-bp bin (locs.rs:321-342). Returns the kept count;
// bit 31: some kept alignment lies in the central region (in_bounds, locs.rs:1008-1014).
__device__ __noinline__ uint32_t gather_sort_dedupe(const Rec16* rec, const uint16_t* nxt, uint32_t head, uint16_t* ord,
                                                    uint32_t boundary, uint32_t contig_len) {
    bool inb = false;
    uint32_t n = 0;
    for (uint32_t v = head; v != NONE16; v = nxt[v]) ord[n++] = static_cast<uint16_t>(v);
    for (uint32_t i = 1; i < n; i++) {
        const uint16_t v = ord[i];
        const double lpv = rec[v].ln_prob;
        uint32_t j = i;
        while (j > 0) {
            const uint16_t u = ord[j - 1];
            const double lpu = rec[u].ln_prob;
            if (lpu < lpv || (lpu == lpv && u > v)) { ord[j] = u; j--; } else break;
        }
        ord[j] = v;
    }
    uint32_t kept = 0;
    for (uint32_t t = 0; t < n; t++) {
        const uint16_t v = ord[t];
        const uint32_t bin = rec[v].start >> 7;                 // STEP_PWR, locs.rs:174,183
        bool dup = false;
        for (uint32_t u = 0; u < kept; u++) dup |= (rec[ord[u]].start >> 7) == bin;
        if (!dup) {
            ord[kept++] = v;
            const uint32_t mid = (rec[v].start + (rec[v].end_rev & ~REV_BIT)) / 2;
            if (boundary <= mid && mid < contig_len - boundary) inb = true;
        }
    }
    return kept | (inb ? 0x80000000u : 0u);
}

struct ContigResult {
    double best;
    uint32_t cnt;
};

// general path: best + kept count of one contig
template <typename C>
__device__ __forceinline__ ContigResult count_pairs(const C& pc, uint32_t max_alns, double prob_diff) {
    double best = -INFINITY;
    enumerate_pairs(pc, [&](double prob, uint32_t, const AlnRef&, const AlnRef&) { best = fmax(best, prob); });
    const double thresh = best - prob_diff;                                  // locs.rs:796
    uint32_t ge = 0;
    enumerate_pairs(pc, [&](double prob, uint32_t, const AlnRef&, const AlnRef&) { ge += prob >= thresh; });
    return ContigResult{best, min(ge, max_alns)};                            // locs.rs:797
}
__device__ __noinline__ ContigResult general_count(const PairCtx pc, uint32_t max_alns, double prob_diff) { return count_pairs(pc, max_alns, prob_diff); }

// general path: selection-emit of the kept PairAlignments (decreasing ln_prob, ties in push order, locs.rs:795)
template <typename C>
__device__ __forceinline__ void emit_pairs(const C& pc, uint32_t cnt, double weight, uint32_t contig, PairAlnDev* out) {
    double prev_prob = INFINITY;
    uint32_t prev_ord = 0;
    bool first = true;
    for (uint32_t e = 0; e < cnt; e++) {
        double bp = -INFINITY; uint32_t bo = NONE32;
        AlnRef b1{0.0, 0, 0, NONE32, false}, b2 = b1;
        enumerate_pairs(pc, [&](double prob, uint32_t ord, const AlnRef& x1, const AlnRef& x2) {
            const bool after = first || prob < prev_prob || (prob == prev_prob && ord > prev_ord);
            if (after && (prob > bp || (prob == bp && ord < bo))) { bp = prob; bo = ord; b1 = x1; b2 = x2; }
        });
        PairAlnDev o;
        o.ln_prob = bp * weight;                                              // locs.rs:861-863
        o.mid1 = b1.idx == NONE32 ? NONE32 : (b1.start + b1.end) / 2;          // Interval::middle, interv.rs:154-156
        o.mid2 = b2.idx == NONE32 ? NONE32 : (b2.start + b2.end) / 2;
        o.contig = static_cast<uint16_t>(contig);
        o.ix1 = b1.idx == NONE32 ? 0xFFFFu : static_cast<uint16_t>(b1.idx);
        o.ix2 = b2.idx == NONE32 ? 0xFFFFu : static_cast<uint16_t>(b2.idx);
        o._pad = 0;
        out[e] = o;
        prev_prob = bp; prev_ord = bo; first = false;
    }
}
__device__ __noinline__ void general_emit(const PairCtx pc, uint32_t cnt, double weight, uint32_t contig, PairAlnDev* out) { emit_pairs(pc, cnt, weight, contig, out); }

// Fast path: at most one saved alignment per read end on this contig. Candidates in push order:
// 0 = (aln1, aln2), 1 = (aln1, unmapped), 2 = (unmapped, aln2).
struct Fast3 {
    double prob[3];
    bool present[3];
    AlnRef a1, a2;
    bool has1, has2;
};

__device__ __forceinline__ Fast3 fast_from_refs(const InsLut& ins, const AlnRef& a1, const AlnRef& a2, bool has1, bool has2,
                                                double unm_ins_penalty, bool paired);
__device__ __forceinline__ Fast3 fast_candidates(const InsLut& ins, const Rec16* rec, uint32_t h1, uint32_t h2, double best0,
                                                 double best1, double unm_ins_penalty, bool paired) {
    const AlnRef none{0.0, 0, 0, NONE32, false};
    const bool has1 = h1 != NONE16, has2 = h2 != NONE16;
    return fast_from_refs(ins, has1 ? load_aln(rec, h1, best0) : none, has2 ? load_aln(rec, h2, best1) : none, has1, has2, unm_ins_penalty, paired);
}
__device__ __forceinline__ Fast3 fast_from_refs(const InsLut& ins, const AlnRef& a1, const AlnRef& a2, bool has1, bool has2,
                                                double unm_ins_penalty, bool paired) {
    Fast3 f;
    f.has1 = has1; f.has2 = has2;
    f.a1 = a1; f.a2 = a2;
    f.present[0] = f.present[1] = f.present[2] = false;
    f.prob[0] = f.prob[1] = f.prob[2] = -INFINITY;
    if (!paired) {                                   // identify_single_end_alignments: one entry (aln, -)
        if (f.has1) { f.present[1] = true; f.prob[1] = f.a1.lp; }
        return f;
    }
    double m = -INFINITY;
    if (f.has1 && f.has2 && f.a1.rev != f.a2.rev) {
        const uint32_t insert = max(f.a1.end, f.a2.end) - min(f.a1.start, f.a2.start);
        const double prob = f.a1.lp + f.a2.lp + ins.ln_prob(insert);
        if (isfinite(prob)) { f.present[0] = true; f.prob[0] = prob; m = prob; }
    }
    if (f.has1) {
        const double alone = f.a1.lp + unm_ins_penalty;
        if (alone >= m) { f.present[1] = true; f.prob[1] = alone; }
    }
    if (f.has2) {
        const double alone = f.a2.lp + unm_ins_penalty;
        if (alone >= m) { f.present[2] = true; f.prob[2] = alone; }
    }
    return f;
}

__device__ __forceinline__ ContigResult fast_count(const Fast3& f, uint32_t max_alns, double prob_diff) {
    const double best = fmax(f.prob[0], fmax(f.prob[1], f.prob[2]));
    const double thresh = best - prob_diff;
    uint32_t ge = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) ge += f.present[i] && f.prob[i] >= thresh;
    return ContigResult{best, min(ge, max_alns)};
}

__device__ __forceinline__ void fast_emit(const Fast3& f, uint32_t cnt, double weight, uint32_t contig, PairAlnDev* out) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
        if (!f.present[i]) continue;
        uint32_t rank = 0;                               // candidates sorted before i (ln_prob desc, push order)
#pragma unroll
        for (int j = 0; j < 3; j++)
            rank += j != i && f.present[j] && (f.prob[j] > f.prob[i] || (f.prob[j] == f.prob[i] && j < i));
        if (rank < cnt) {
            const bool use1 = i != 2, use2 = i != 1;
            PairAlnDev o;
            o.ln_prob = f.prob[i] * weight;
            o.mid1 = use1 ? (f.a1.start + f.a1.end) / 2 : NONE32;
            o.mid2 = use2 ? (f.a2.start + f.a2.end) / 2 : NONE32;
            o.contig = static_cast<uint16_t>(contig);
            o.ix1 = use1 ? static_cast<uint16_t>(f.a1.idx) : 0xFFFFu;
            o.ix2 = use2 ? static_cast<uint16_t>(f.a2.idx) : 0xFFFFu;
            o._pad = 0;
            out[rank] = o;
        }
    }
}

// ContigInfo::read_end_weight (windows.rs:493-503): the largest explicit weight among the middle of the read end and half a
// window to either side; ExplicitWeights holds len + 1 values per allele (finish(), 221-224). None -> 0.0.
__device__ __forceinline__ double read_end_weight(const LocusView& L, uint32_t contig, uint32_t clen, uint32_t middle) {
    if (middle == NONE32) return 0.0;
    const double* val = L.ew_val + L.ew_off[contig];
    const uint32_t u = L.half_window;
    const uint32_t lo = middle > u ? middle - u : 0u;
    const uint32_t hi = min(middle + u, clen);                  // n - 1 with n = len + 1
    return fmax(fmax(val[middle], val[lo]), val[hi]);
}

// barrier between the passes over a pair; parked alignments in global memory (BIG) are made visible to the other lanes first
template <bool BIG>
__device__ __forceinline__ void pair_barrier() {
    if constexpr (BIG) __threadfence_block();
    __syncthreads();
}

// EW = explicit region weights are set (lcty_locus_set_explicit_weights): the pair's weight is multiplied by
// ContigInfos::explicit_read_weight over its PairAlignments (windows.rs:683-693; locs.rs:860, 903) before it scales them.
// BIG = the saved alignments of a pair (16 B each + two 16-bit links) do not fit the LDS next to the per-allele tables (4 096
// alleles with an alignment per read end on each: 209 KB): they are parked in a scratch of the workgroup in global memory
// (L2-resident: written and read back by the same CU within microseconds); the per-allele tables stay in LDS.
// CNT = the batch holds counted alignments (lcty_reads_append_counted) instead of BAM records with CIGAR words.
template <bool EW, bool BIG, bool CNT = false>
__device__ __forceinline__ void score_reads_body(const LocusView& L, const ReadsView& R, const uint32_t max_recs) {
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t A = L.n_alleles;
    const uint32_t mr2 = (max_recs + 1) & ~1u;
    // region sizes must match score_lds_bytes() / score_park_bytes()
    uint8_t* const park_base = BIG ? R.park + static_cast<uint64_t>(blockIdx.x) * R.park_stride : smem;
    Rec16* rec = reinterpret_cast<Rec16*>(park_base);                         // [max_recs] saved alignments, by record index
    uint16_t* nxt = reinterpret_cast<uint16_t*>(rec + max_recs);              // [max_recs] chain links
    uint16_t* order = nxt + mr2;                                              // [max_recs] scratch of the general path
    // LDS form: [rec | nxt | order | head32 | alen | cursor | kk1 kk2 cnt8]; BIG: LDS holds head32 + cursor only (4 B per allele, so
    // that ten workgroups fit a CU at 4 096 alleles), the lane-private per-allele bytes go to the park, allele lengths come from L2
    uint8_t* const tables = BIG ? smem : reinterpret_cast<uint8_t*>(order + mr2);
    uint32_t* head32 = reinterpret_cast<uint32_t*>(tables);                   // [A] {head(end 0) | head(end 1) << 16}
    uint32_t* alen_lds = head32 + A;                                          // [A] allele lengths (filled once per workgroup; not BIG)
    uint32_t* scratch_cursor = BIG ? head32 + A : alen_lds + A;               // [1]
    uint8_t* kk1 = BIG ? reinterpret_cast<uint8_t*>(order + mr2) : reinterpret_cast<uint8_t*>(scratch_cursor + 1);   // [A] general path: kept first-end alns (<= 10)
    uint8_t* kk2 = kk1 + A;                                                   // [A]
    uint8_t* cnt8 = kk2 + A;                                                  // [A] emitted PairAlignments (<= 10); bit 7 = general path
    // SPLIT (counted alignments in LDS): the two chain heads of a contig in two words — head(end 0) in head32[c], head(end 1) in
    // head32[A + c], where the record form keeps the allele lengths (a counted alignment has its clipping applied: pass 1 does not
    // need them, pass 3 reads them from L2 along the contigs) — so that chaining a saved alignment is ONE LDS exchange instead of a
    // read and a compare-and-swap loop on a shared word.
    constexpr bool SPLIT = CNT && !BIG;
    const uint32_t* alen = (BIG || SPLIT) ? L.allele_len : alen_lds;
    const int lane = threadIdx.x;
    const bool paired = L.is_paired != 0;
    const InsLut ins{L.ins_lut, L.ins_lut_size, L.ins_n, L.ins_lnq, L.ins_lnpmf_const};
    if constexpr (!BIG && !SPLIT) for (uint32_t i = lane; i < A; i += WAVE) alen_lds[i] = L.allele_len[i];
    __syncthreads();

    unsigned long long pool_at = 0, pool_left = 0;          // this wavefront's share of the pair-alignment arena
    // all pairs of the batch, or the ones the lean kernel left (their indices and their number are on the device)
    const uint64_t n_todo = R.only_list ? static_cast<uint64_t>(*R.only_count) : R.n_pairs;
    for (uint64_t todo = blockIdx.x; todo < n_todo; todo += gridDim.x) {
        const uint64_t p = R.only_list ? R.only_list[todo] : todo;
        const uint64_t a0 = R.aln_off[p];
        const uint32_t* cig = CNT ? nullptr : R.cigar + R.cigar_off[p];
        const uint2 meta = R.pair_meta[p];                 // {index of the mate-2 primary (or n), records to look at}
        const uint32_t j2 = meta.x, n_eff = meta.y;
        const uint32_t split = min(j2, n_eff);
        const uint32_t len0 = R.mate_len[2 * p], len1 = paired ? R.mate_len[2 * p + 1] : 0u;
        double* mrow = R.matrix + p * A;
        const lcty_aln_rec* recs = R.recs + a0;

        // ---------------- prologue: prefetch read bases (consumed by K2) ----------------
        const bool regs_ok = len0 <= 2016 && len1 <= 2016 && L.k <= 31;      // k-mers of 32..63 bases take the general path
        uint64_t bw0 = 0, bw1 = 0;
        uint32_t nm0 = 0, nm1 = 0;
        if (regs_ok) {
            const uint64_t off0 = R.mate_off[2 * p], off1 = R.mate_off[2 * p + 1];
            if (static_cast<uint32_t>(lane) * 32u < len0) {
                bw0 = (reinterpret_cast<const uint64_t*>(R.bases2) + (off0 >> 5))[lane];
                nm0 = (R.nmask + (off0 >> 5))[lane];
            }
            if (static_cast<uint32_t>(lane) * 32u < len1) {
                bw1 = (reinterpret_cast<const uint64_t*>(R.bases2) + (off1 >> 5))[lane];
                nm1 = (R.nmask + (off1 >> 5))[lane];
            }
        }
        for (uint32_t i = lane; i < (SPLIT ? 2 * A : A); i += WAVE) head32[i] = 0xFFFFFFFFu;
        if (lane == 0) *scratch_cursor = 0;
        // the records of the first group of pass 1 are requested now, so that they arrive while lanes 0 / 1 work out the
        // thresholds (two more dependent loads each)
        uint4 raw_first[GR];
#pragma unroll
        for (int g = 0; g < GR; g++) {
            const uint32_t idx = g * WAVE + lane;
            raw_first[g] = idx < n_eff ? reinterpret_cast<const uint4*>(recs)[idx] : make_uint4(0, 0, 0, 0);
        }

        // ---------------- thresholds (lanes 0 / 1 score the primaries) ----------------
        // state: 1 examined-ok, 2 primary saved, 4 error
        uint32_t my_good = 0, my_thr = NONE32, my_pass = NONE32, my_state = 0;
        if (lane < 2) {
            const uint32_t e = lane;
            const uint32_t pidx = e ? j2 : 0u;
            const bool exists = e == 0 ? n_eff > 0 : (paired && j2 < n_eff);
            if (e == 1 && !paired) {
                my_state = 3;                                   // single-end: second end is vacuously fine
            } else if (!exists) {
                my_state = 4;                                   // expect("Cannot read any more records"), locs.rs:509
            } else {
                const uint4 praw = reinterpret_cast<const uint4*>(recs)[pidx];
                const lcty_aln_rec pr = recs[pidx];
                const uint32_t read_len = e ? len1 : len0;
                const uint32_t pflags = CNT ? ((praw.x >> 28 & 1u) ? LCTY_FLAG_REVERSE : 0u) | ((praw.x >> 29 & 1u) ? LCTY_FLAG_SECONDARY : 0u)
                                              | ((praw.x >> 30 & 1u) ? LCTY_FLAG_UNMAPPED : 0u) : pr.flags;
                const uint32_t pcontig = CNT ? (praw.y & 0xFFFFu) : pr.contig;
                const bool is_primary = (pflags & (LCTY_FLAG_SECONDARY | LCTY_FLAG_SUPPL)) == 0;
                if (read_len == 0 || !is_primary) my_state = 4;                 // locs.rs:511-517 / LaggedReader assert
                else if (pflags & LCTY_FLAG_UNMAPPED) my_state = 0;             // locs.rs:520-523
                else if ((!CNT && pr.n_cigar == 0) || pcontig >= A) my_state = 4;
                else {
                    Scored sc;
                    if constexpr (CNT) sc = score_counted(L, praw);
                    else {
                        const W8 w8 = *reinterpret_cast<const W8*>(cig + pr.cigar_rel);
                        const OpCounts oc = count_ops(w8, cig + pr.cigar_rel, pr.n_cigar, true);
                        sc = score_counts(L, oc, pr.pos, L.allele_len[pr.contig]);
                    }
                    if (sc.bad) my_state = 4;
                    else {
                        const uint2 gp = L.edit_lut[min(read_len, L.edit_lut_size - 1)];
                        uint32_t good = gp.x, passable = gp.y, thr = good;
                        double compl_v = 1.0;
                        if (L.short_reads) {                     // neighb_complexity, windows.rs:447-452, 696-698
                            const uint32_t mid = (sc.start + sc.end) / 2;
                            const uint32_t o = L.ci_off[pcontig];
                            const uint32_t npos = L.ci_off[pcontig + 1] - o;
                            const uint32_t i = min(mid > L.half_neighb ? mid - L.half_neighb : 0u, npos - 1);
                            compl_v = static_cast<double>(L.compl_cnt[o + i]) * L.compl_mult;
                        }
                        if (compl_v <= L.poor_compl) {           // locs.rs:533-536
                            thr = max(good, static_cast<uint32_t>(L.poor_compl_edit * static_cast<double>(read_len)));
                            passable += thr - good;
                        }
                        my_good = good; my_thr = thr; my_pass = passable;
                        my_state = 1u | (sc.edit <= passable ? 2u : 0u);
                    }
                }
            }
        }
        const uint32_t good0 = __shfl(my_good, 0), thr0 = __shfl(my_thr, 0), pass0 = __shfl(my_pass, 0), st0 = __shfl(my_state, 0);
        const uint32_t good1 = __shfl(my_good, 1), thr1 = __shfl(my_thr, 1), pass1 = __shfl(my_pass, 1), st1 = __shfl(my_state, 1);
        pair_barrier<BIG>();      // head table initialised

        // ---------------- pass 1: records -> (best edit, best ln-prob), saved ones -> LDS chains ----------------
        uint32_t be0 = NONE32, be1 = NONE32, bad0 = 0, bad1 = 0;
        double bl0 = -INFINITY, bl1 = -INFINITY;
        // a pair whose first primary was not pushed is rejected without looking further (locs.rs:539-543)
        const bool look = (st0 & 3u) == 3u;
        for (uint32_t base = 0; look && base < n_eff; base += WAVE * GR) {
            uint4 raw[GR];
            W8 cw[GR];
            uint32_t clen[GR];
#pragma unroll
            for (int g = 0; g < GR; g++) {
                const uint32_t idx = base + g * WAVE + lane;
                if (base == 0) raw[g] = raw_first[g];
                else raw[g] = idx < n_eff ? reinterpret_cast<const uint4*>(recs)[idx] : make_uint4(0, 0, 0, 0);
            }
            if constexpr (!CNT) {
#pragma unroll
                for (int g = 0; g < GR; g++) {
                    if (raw[g].z) cw[g] = *reinterpret_cast<const W8*>(cig + raw[g].w);
                    else {
#pragma unroll
                        for (int i = 0; i < 8; i++) cw[g].w[i] = 0;
                    }
                    const uint32_t contig = raw[g].y & 0xFFFFu;
                    clen[g] = contig < A ? alen[contig] : 0u;
                }
            }
#pragma unroll
            for (int g = 0; g < GR; g++) {
                const uint32_t idx = base + g * WAVE + lane;
                const uint32_t nc = CNT ? (((raw[g].x >> 30) & 1u) ? 0u : 1u) : raw[g].z;      // counted: an unmapped record has no alignment (an empty CIGAR in the record form)
                const uint32_t pos = CNT ? (raw[g].x & 0x0FFFFFFFu) : raw[g].x, contig = raw[g].y & 0xFFFFu;
                const uint32_t bflags = CNT ? (((raw[g].x >> 28) & 1u) ? LCTY_FLAG_REVERSE : 0u) | (((raw[g].x >> 30) & 1u) ? LCTY_FLAG_UNMAPPED : 0u)
                                            : raw[g].y >> 16;
                const bool primary = idx == 0 || idx == j2;
                // empty CIGAR: skipped with a warning (locs.rs:550-554); unmapped primaries end the pair above
                if (idx < n_eff && nc != 0 && !(primary && (bflags & LCTY_FLAG_UNMAPPED))) {
                    const uint32_t e = idx >= split ? 1u : 0u;
                    const bool cbad = contig >= A;
                    Scored sc;
                    if constexpr (CNT) sc = score_counted(L, raw[g]);
                    else {
                        const OpCounts oc = count_ops(cw[g], cig + raw[g].w, nc, primary);
                        sc = score_counts(L, oc, pos, clen[g]);
                    }
                    (void)pos;
                    if (sc.bad || cbad) {
                        if (!primary) { if (e) bad1 = 1; else bad0 = 1; }   // primaries are judged by lanes 0/1 above
                    } else {
                        if (e == 0) { be0 = min(be0, sc.edit); bl0 = fmax(bl0, sc.ln_prob); }
                        else { be1 = min(be1, sc.edit); bl1 = fmax(bl1, sc.ln_prob); }
                        if (sc.edit <= (e ? pass1 : pass0)) {              // save (locs.rs:314)
                            Rec16 r;
                            r.ln_prob = sc.ln_prob; r.start = sc.start;
                            r.end_rev = sc.end | ((bflags & LCTY_FLAG_REVERSE) ? REV_BIT : 0u);
                            rec[idx] = r;
                            // chain per (contig, end): exchange the 16-bit head inside its 32-bit word
                            if constexpr (SPLIT) {
                                nxt[idx] = static_cast<uint16_t>(atomicExch(&head32[e * A + contig], 0xFFFF0000u | idx));
                            } else {
                                uint32_t* word = &head32[contig];
                                uint32_t old = *word, assumed;
                                do {
                                    assumed = old;
                                    const uint32_t repl = e ? ((assumed & 0x0000FFFFu) | (idx << 16)) : ((assumed & 0xFFFF0000u) | idx);
                                    old = atomicCAS(word, assumed, repl);
                                } while (old != assumed);
                                nxt[idx] = static_cast<uint16_t>(e ? (old >> 16) : (old & 0xFFFFu));
                            }
                        }
                    }
                }
            }
        }
        be0 = wave_min_u32(be0); be1 = wave_min_u32(be1);
        bl0 = wave_max_f64(bl0); bl1 = wave_max_f64(bl1);
        bad0 = wave_sum_u32(bad0); bad1 = wave_sum_u32(bad1);
        pair_barrier<BIG>();

        // end 1 is only looked at when end 0 is well mapped (locs.rs:1125-1132); its secondaries only when its
        // primary was pushed
        const bool wm0 = look && be0 <= (L.strict_subset ? pass0 : thr0) && !bad0;
        const bool saved1 = (st1 & 3u) == 3u;
        const bool err = (st0 & 4u) || (look && bad0) || (wm0 && ((st1 & 4u) || (saved1 && bad1)));
        if (err && lane == 0) atomicMax(R.err_flag, static_cast<uint32_t>(LCTY_ERR_INVALID_DATA));
        const bool wm1 = !paired || (saved1 && be1 <= (L.strict_subset ? pass1 : thr1));
        bool accepted = wm0 && wm1 && !err;
        double weight = 1.0;
        if (accepted) {
            weight *= be0 <= good0 ? 1.0 : sqrt(static_cast<double>(good0) / static_cast<double>(be0));      // locs.rs:565
            if (paired) weight *= be1 <= good1 ? 1.0 : sqrt(static_cast<double>(good1) / static_cast<double>(be1));
        }
        uint8_t status = LCTY_READ_POORLY_MAPPED;
        uint32_t total_cnt = 0, uk0 = 0, uk1 = 0;
        double unmapped_prob = 0.0;
        uint64_t pa_base = 0;

        if (!accepted && lane == 0) R.recover_w[p] = -1.0;
        if (accepted) {
            // ---------------- K2: unique k-mers -> read weight (locs.rs:968-1002) ----------------
            // (evaluated after the in-bounds test in the reference; no side effects, order irrelevant)
            if (regs_ok) pair_unique_kmers_regs(L.kset, L.kset_mask, L.undef_in_set, L.k, len0, len1, bw0, bw1, nm0, nm1, lane, &uk0, &uk1);
            else {
                const uint64_t off0 = R.mate_off[2 * p], off1 = R.mate_off[2 * p + 1];
                uk0 = mate_unique_kmers_global(L.kset, L.kset_mask, L.undef_in_set, L.k,
                                               reinterpret_cast<const uint64_t*>(R.bases2) + (off0 >> 5), R.nmask + (off0 >> 5), len0, lane);
                uk1 = len1 ? mate_unique_kmers_global(L.kset, L.kset_mask, L.undef_in_set, L.k,
                                                      reinterpret_cast<const uint64_t*>(R.bases2) + (off1 >> 5), R.nmask + (off1 >> 5), len1, lane)
                           : 0u;
            }
            const uint32_t paired_count = (uk0 + uk1) & 0xFFFFu;
            double kw = L.weight_interc + static_cast<double>(paired_count) * L.weight_mult;
            kw = kw < 0.0 ? 0.0 : (kw > 1.0 ? 1.0 : kw);
            weight *= kw;
            const uint32_t max_alns = weight >= L.min_weight ? LCTY_MAX_USED_ALNS : LCTY_MAX_UNUSED_ALNS;   // locs.rs:1268
            const double unm_ins_penalty = L.unmapped_penalty + L.insert_penalty;                         // locs.rs:816
            unmapped_prob = paired ? weight * (2.0 * L.unmapped_penalty + L.insert_penalty)                // locs.rs:866
                                   : weight * L.unmapped_penalty;                                          // locs.rs:908

            // ---------------- pass 3a: per contig best + kept count, matrix row ----------------
            bool inb = false;
            for (uint32_t c0 = 0; c0 < A; c0 += WAVE) {
                const uint32_t c = c0 + lane;
                uint32_t h1 = NONE16, h2 = NONE16;
                bool general = false;
                if (c < A) {
                    if constexpr (SPLIT) { h1 = head32[c] & 0xFFFFu; h2 = head32[A + c] & 0xFFFFu; }
                    else { const uint32_t hw = head32[c]; h1 = hw & 0xFFFFu; h2 = hw >> 16; }
                    general = (h1 != NONE16 && nxt[h1] != NONE16) || (h2 != NONE16 && nxt[h2] != NONE16);
                }
                ContigResult res{-INFINITY, 0};
                if (c < A && !general) {
                    const Fast3 f = fast_candidates(ins, rec, h1, h2, bl0, bl1, unm_ins_penalty, paired);
                    const uint32_t clen = alen[c];
                    if (f.has1) { const uint32_t mid = (f.a1.start + f.a1.end) / 2; inb |= L.boundary <= mid && mid < clen - L.boundary; }
                    if (f.has2) { const uint32_t mid = (f.a2.start + f.a2.end) / 2; inb |= L.boundary <= mid && mid < clen - L.boundary; }
                    if (f.has1 || f.has2) res = fast_count(f, max_alns, L.prob_diff);
                    cnt8[c] = static_cast<uint8_t>(res.cnt);
                }
                if (__ballot(general)) {                  // some lane needs the general path (wave-uniform branch)
                    if (general) {
                        uint32_t n1 = 0, n2 = 0;
                        for (uint32_t v = h1; v != NONE16; v = nxt[v]) n1++;
                        for (uint32_t v = h2; v != NONE16; v = nxt[v]) n2++;
                        const uint32_t o = atomicAdd(scratch_cursor, n1 + n2);
                        const uint32_t clen = alen[c];
                        const uint32_t K1 = gather_sort_dedupe(rec, nxt, h1, order + o, L.boundary, clen);
                        const uint32_t K2 = gather_sort_dedupe(rec, nxt, h2, order + o + n1, L.boundary, clen);
                        inb |= ((K1 | K2) & 0x80000000u) != 0;
                        const uint32_t k1 = min(K1 & 0x7FFFFFFFu, max_alns), k2 = min(K2 & 0x7FFFFFFFu, max_alns);   // locs.rs:842-851
                        kk1[c] = static_cast<uint8_t>(k1); kk2[c] = static_cast<uint8_t>(k2);
                        // chains are consumed: {start of list 2, start of list 1} in `order`
                        if constexpr (SPLIT) { head32[c] = o + n1; head32[A + c] = o; }
                        else head32[c] = (o + n1) | (o << 16);
                        const PairCtx pc{ins, rec, order + o, order + o + n1, k1, k2, bl0, bl1, unm_ins_penalty, paired};
                        res = general_count(pc, max_alns, L.prob_diff);
                        cnt8[c] = static_cast<uint8_t>(res.cnt | 0x80u);
                    }
                }
                if (c < A) {
                    // locs.rs:861-863, 621-629 (EW: the weight is not final yet; scaled below, by the lane that wrote it)
                    mrow[c] = res.cnt ? (EW ? res.best : res.best * weight) : unmapped_prob;
                    total_cnt += res.cnt;
                }
            }
            const bool any_inb = __ballot(inb) != 0ull;
            total_cnt = wave_sum_u32(total_cnt);
            if (lane == 0) R.recover_w[p] = any_inb ? weight : -1.0;                         // alignment recovery looks at these pairs
            const bool edit_good = be0 <= thr0 && (!paired || be1 <= thr1);                   // best_edit_is_good, locs.rs:293-295
            if (!any_inb) { status = LCTY_READ_OUT_OF_BOUNDS; accepted = false; }
            else if (!edit_good) { status = LCTY_READ_POORLY_MAPPED; accepted = false; }
            else status = weight >= L.min_weight ? LCTY_READ_GOOD : LCTY_READ_FEW_KMERS;     // locs.rs:1277-1285 (EW: again below)
            pair_barrier<BIG>();

            if (accepted) {
                // The arena cursor is ONE word in memory: an atomic per read pair from every wavefront of the device queues up at its
                // L2 channel (2.4 of the kernel's 17.4 ms at 1 M pairs). A wavefront takes PA_CHUNK entries at a time and hands
                // them to its pairs (launches with at least PA_POOL_MIN_PAIRS pairs per wavefront; ReadsView::pa_chunk); the arena is
                // addressed through pa_off / pa_cnt only, and lcty_reads_create adds the room (an eighth + a chunk per wavefront).
                if (R.pa_chunk == 0 || static_cast<uint64_t>(total_cnt) * 8 > R.pa_chunk) {
                    // small launches and unusually large pairs: their own reservation (nothing of a chunk is left unused for them)
                    if (lane == 0) pa_base = atomicAdd(R.pa_count, static_cast<unsigned long long>(total_cnt));
                    pa_base = __shfl(pa_base, 0);
                } else {
                    if (total_cnt > pool_left) {                                  // what is left (< an eighth of a chunk) stays unused
                        if (lane == 0) pool_at = atomicAdd(R.pa_count, static_cast<unsigned long long>(R.pa_chunk));
                        pool_at = __shfl(pool_at, 0);
                        pool_left = R.pa_chunk;
                    }
                    pa_base = pool_at;
                    pool_at += total_cnt; pool_left -= total_cnt;
                }
                if (lane == 0 && pa_base + total_cnt > R.pa_cap) atomicMax(R.err_flag, static_cast<uint32_t>(LCTY_ERR_RUNTIME));
                const bool room = pa_base + total_cnt <= R.pa_cap;
                // ---------------- pass 3b: emit PairAlignments, contig-ascending ----------------
                const double emit_weight = EW ? 1.0 : weight;
                double ew_sum = 0.0;
                uint32_t run = 0;
                for (uint32_t c0 = 0; c0 < A; c0 += WAVE) {
                    const uint32_t c = c0 + lane;
                    const uint32_t craw = c < A ? cnt8[c] : 0u;
                    const uint32_t cnt = craw & 0x7Fu;
                    const bool general = (craw & 0x80u) != 0;
                    uint32_t tot;
                    const uint32_t my_off = run + wave_excl_scan_u32(cnt, lane, &tot);
                    run += tot;
                    PairAlnDev* out = R.pa + pa_base + my_off;
                    if (c < A) R.pa_idx[p * A + c] = my_off | (cnt << 24);      // direct (pair, contig) -> entries index
                    if (room && cnt && !general) {
                        const uint32_t hw = SPLIT ? (head32[c] & 0xFFFFu) | (head32[A + c] << 16) : head32[c];
                        const Fast3 f = fast_candidates(ins, rec, hw & 0xFFFFu, hw >> 16, bl0, bl1, unm_ins_penalty, paired);
                        fast_emit(f, cnt, emit_weight, c, out);
                    }
                    if (__ballot(room && general && cnt)) {
                        if (room && general && cnt) {
                            const uint32_t hw = SPLIT ? (head32[c] & 0xFFFFu) | (head32[A + c] << 16) : head32[c];
                            const PairCtx pc{ins, rec, order + (hw >> 16), order + (hw & 0xFFFFu), kk1[c], kk2[c], bl0, bl1,
                                             unm_ins_penalty, paired};
                            general_emit(pc, cnt, emit_weight, c, out);
                        }
                    }
                    if constexpr (EW) {
                        // every lane reads back the entries it has just written itself (same thread: program order)
                        if (room && cnt) {
                            const uint32_t clen = alen[c];
                            for (uint32_t e = 0; e < cnt; e++)
                                ew_sum += fmax(read_end_weight(L, c, clen, out[e].mid1), read_end_weight(L, c, clen, out[e].mid2));
                        }
                    }
                }
                if constexpr (EW) {
                    // explicit_read_weight = mean over all PairAlignments of the pair (windows.rs:688-692); the sum is taken
                    // per lane and then across the wave (a different order of additions than upstream's serial loop)
                    ew_sum = wave_sum_f64(ew_sum);
                    weight = weight * (ew_sum / static_cast<double>(total_cnt));                       // locs.rs:860, 903
                    unmapped_prob = paired ? weight * (2.0 * L.unmapped_penalty + L.insert_penalty) : weight * L.unmapped_penalty;
                    status = weight >= L.min_weight ? LCTY_READ_GOOD : LCTY_READ_FEW_KMERS;           // GrouppedAlignments::weight()
                    for (uint32_t c = lane; c < A; c += WAVE) {
                        const uint32_t ix = R.pa_idx[p * A + c];
                        const uint32_t cnt = ix >> 24;
                        mrow[c] = cnt ? mrow[c] * weight : unmapped_prob;
                        if (room) {
                            PairAlnDev* out = R.pa + pa_base + (ix & 0xFFFFFFu);
                            for (uint32_t e = 0; e < cnt; e++) out[e].ln_prob = out[e].ln_prob * weight;
                        }
                    }
                }
            }
        }

        // rows of pairs that are not in AllAlignments::reads contribute 0.0 to every genotype score
        if (status != LCTY_READ_GOOD)
            for (uint32_t c = lane; c < A; c += WAVE) mrow[c] = 0.0;
        if (lane == 0) {
            R.status[p] = status;
            R.weight[p] = accepted ? weight : 0.0;
            R.unmapped_prob[p] = accepted ? unmapped_prob : 0.0;
            R.pa_off[p] = accepted ? pa_base : 0ull;
            R.pa_cnt[p] = accepted ? total_cnt : 0u;
            R.uniq_kmers[2 * p] = accepted ? static_cast<uint16_t>(uk0) : 0;
            R.uniq_kmers[2 * p + 1] = accepted ? static_cast<uint16_t>(uk1) : 0;
        }
        pair_barrier<BIG>();
    }
}

// ---------------------------------------------------------------------------------------------
// The LEAN form for the common case of counted batches: every (contig, read end) of a pair has at most ONE saved alignment.
// Then nothing has to be chained, ordered or parked: pass 1 leaves, per (contig, end), the INDEX of the saved record (one LDS
// exchange; an exchange that finds an index there marks the pair), and pass 3 reads that 16-byte record again — it is in the L2,
// the pair's records were read microseconds ago — and scores it once more instead of keeping 16 bytes per record in LDS. LDS per
// wavefront: 8 B per allele + a byte per allele (13 KB -> 2.3 KB at 256 alleles) and half the registers: the kernel is bound by
// the latency of its dependent steps (67 % of the wavefront cycles waiting), so wavefronts in flight are what it needs.
// A marked pair — or one whose mates are too long for the register path of the k-mer windows — is left untouched and appended
// to `defer_list`; the general kernel scores those afterwards. Same arithmetic, same order per pair: the products are the
// general kernel's bit for bit (tests/test_gpu_counted.py runs both).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ AlnRef ref_from_counted(const LocusView& L, const uint4 raw, uint32_t idx, double best) {
    const Scored sc = score_counted(L, raw);
    return AlnRef{sc.ln_prob - best, sc.start, sc.end, idx, ((raw.x >> 28) & 1u) != 0};            // normalize_probs, locs.rs:358-360
}

constexpr uint32_t LEAN_OVF = 64;             // second saved alignments of (contig, read end) groups a pair of the TWO form can hold
template <bool TIMED, bool KEEP, bool TWO = false>
__device__ __forceinline__ void score_lean_body(const LocusView& L, const ReadsView& R) {
    static_assert(!TWO || KEEP, "the two-slot form keeps the saved alignments' products in LDS");
    extern __shared__ __align__(16) uint8_t smem[];
    // TIMED (a diagnostic build of the same code, lcty_ctx_set_knob "score_timing"): where a wavefront's time goes, phase by phase
    uint64_t tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0;
    auto stamp = [&](int k) {
        if constexpr (TIMED) {
            __builtin_amdgcn_sched_barrier(0);
            const uint64_t t = __builtin_amdgcn_s_memtime();
            tph[k] += t - t_prev; t_prev = t;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if constexpr (TIMED) t_prev = __builtin_amdgcn_s_memtime();
    const uint32_t A = L.n_alleles;
    // KEEP (few alleles: 33 B of LDS per allele): what pass 1 computed of the saved alignment — its place, strand and index packed
    // into 64 bits, its ln-probability — stays in LDS and pass 3 has it from there; otherwise (9 B per allele) pass 1 leaves the
    // record's index and pass 3 reads and scores the record again.
    uint32_t* head = reinterpret_cast<uint32_t*>(smem);                       // [2A]: saved record of (contig, end 0) | (contig, end 1)
    unsigned long long* kept = reinterpret_cast<unsigned long long*>(smem);   // KEEP [2A]: start | rev << 28 | (end - start) << 29 | index << 47
    double* kept_lp = reinterpret_cast<double*>(kept + 2 * A);                // KEEP [2A]
    uint8_t* cnt8 = KEEP ? reinterpret_cast<uint8_t*>(kept_lp + 2 * A) : reinterpret_cast<uint8_t*>(head + 2 * A);   // [A] PairAlignments of the contig (<= 3)
    // TWO: the second saved alignment of a (contig, read end) group, where there is one: a list of [LEAN_OVF] packed values with its count,
    // and per group the place of its second on that list ([2A], LEAN_NONE: none) — an exchange on it tells a third alignment of a group
    unsigned long long* ovf_val = reinterpret_cast<unsigned long long*>(smem + ((static_cast<size_t>(A) * 33 + 7) & ~static_cast<size_t>(7)));
    uint32_t* ovf_n = reinterpret_cast<uint32_t*>(ovf_val + LEAN_OVF);
    uint32_t* ovf_at = ovf_n + 2;
    constexpr uint32_t LEAN_NONE = 0xFFFFFFFFu;
    constexpr unsigned long long NOT_KEPT = ~0ull;
    const int lane = threadIdx.x;
    const bool paired = L.is_paired != 0;
    const InsLut ins{L.ins_lut, L.ins_lut_size, L.ins_n, L.ins_lnq, L.ins_lnpmf_const};
    unsigned long long pool_at = 0, pool_left = 0;
    // TWO: the pairs the first lean launch left (only_list); what this form cannot take either goes on to defer_list
    const uint64_t n_todo = TWO ? static_cast<uint64_t>(*R.only_count) : R.n_pairs;
    for (uint64_t todo = blockIdx.x; todo < n_todo; todo += gridDim.x) {
        const uint64_t p = TWO ? R.only_list[todo] : todo;
        const uint64_t a0 = R.aln_off[p];
        const uint2 meta = R.pair_meta[p];
        const uint32_t j2 = meta.x, n_eff = meta.y;
        const uint32_t split = min(j2, n_eff);
        const uint32_t len0 = R.mate_len[2 * p], len1 = paired ? R.mate_len[2 * p + 1] : 0u;
        double* mrow = R.matrix + p * A;
        const uint4* recs = reinterpret_cast<const uint4*>(R.recs + a0);
        const bool regs_ok = len0 <= 2016 && len1 <= 2016;
        uint64_t bw0 = 0, bw1 = 0;
        uint32_t nm0 = 0, nm1 = 0;
        if (regs_ok) {
            const uint64_t off0 = R.mate_off[2 * p], off1 = R.mate_off[2 * p + 1];
            if (static_cast<uint32_t>(lane) * 32u < len0) {
                bw0 = (reinterpret_cast<const uint64_t*>(R.bases2) + (off0 >> 5))[lane];
                nm0 = (R.nmask + (off0 >> 5))[lane];
            }
            if (static_cast<uint32_t>(lane) * 32u < len1) {
                bw1 = (reinterpret_cast<const uint64_t*>(R.bases2) + (off1 >> 5))[lane];
                nm1 = (R.nmask + (off1 >> 5))[lane];
            }
        }
        if constexpr (KEEP) { for (uint32_t i = lane; i < 2 * A; i += WAVE) { kept[i] = NOT_KEPT; if (TWO) ovf_at[i] = LEAN_NONE; } if (TWO && lane == 0) *ovf_n = 0; }
        else { for (uint32_t i = lane; i < 2 * A; i += WAVE) head[i] = 0xFFFFFFFFu; }
        uint4 raw_first[GR];
#pragma unroll
        for (int g = 0; g < GR; g++) {
            const uint32_t idx = g * WAVE + lane;
            raw_first[g] = idx < n_eff ? recs[idx] : make_uint4(0, 0, 0, 0);
        }
        // ---------------- thresholds (lanes 0 / 1 score the primaries): as in the general kernel ----------------
        uint32_t my_good = 0, my_thr = NONE32, my_pass = NONE32, my_state = 0;
        if (lane < 2) {
            const uint32_t e = lane;
            const uint32_t pidx = e ? j2 : 0u;
            const bool exists = e == 0 ? n_eff > 0 : (paired && j2 < n_eff);
            if (e == 1 && !paired) my_state = 3;
            else if (!exists) my_state = 4;
            else {
                const uint4 praw = recs[pidx];
                const uint32_t read_len = e ? len1 : len0;
                const uint32_t pcontig = praw.y & 0xFFFFu;
                const bool is_primary = ((praw.x >> 29) & 1u) == 0;
                if (read_len == 0 || !is_primary) my_state = 4;
                else if ((praw.x >> 30) & 1u) my_state = 0;
                else if (pcontig >= A) my_state = 4;
                else {
                    const Scored sc = score_counted(L, praw);
                    const uint2 gp = L.edit_lut[min(read_len, L.edit_lut_size - 1)];
                    uint32_t good = gp.x, passable = gp.y, thr = good;
                    double compl_v = 1.0;
                    if (L.short_reads) {
                        const uint32_t mid = (sc.start + sc.end) / 2;
                        const uint32_t o = L.ci_off[pcontig];
                        const uint32_t npos = L.ci_off[pcontig + 1] - o;
                        const uint32_t i = min(mid > L.half_neighb ? mid - L.half_neighb : 0u, npos - 1);
                        compl_v = static_cast<double>(L.compl_cnt[o + i]) * L.compl_mult;
                    }
                    if (compl_v <= L.poor_compl) {
                        thr = max(good, static_cast<uint32_t>(L.poor_compl_edit * static_cast<double>(read_len)));
                        passable += thr - good;
                    }
                    my_good = good; my_thr = thr; my_pass = passable;
                    my_state = 1u | (sc.edit <= passable ? 2u : 0u);
                }
            }
        }
        const uint32_t good0 = __shfl(my_good, 0), thr0 = __shfl(my_thr, 0), pass0 = __shfl(my_pass, 0), st0 = __shfl(my_state, 0);
        const uint32_t good1 = __shfl(my_good, 1), thr1 = __shfl(my_thr, 1), pass1 = __shfl(my_pass, 1), st1 = __shfl(my_state, 1);
        __syncthreads();          // head table initialised
        stamp(0);

        // ---------------- pass 1 ----------------
        uint32_t be0 = NONE32, be1 = NONE32, bad0 = 0, bad1 = 0;
        double bl0 = -INFINITY, bl1 = -INFINITY;
        bool multi = false;
        const bool look = (st0 & 3u) == 3u;
        for (uint32_t base = 0; look && base < n_eff; base += WAVE * GR) {
            uint4 raw[GR];
#pragma unroll
            for (int g = 0; g < GR; g++) {
                const uint32_t idx = base + g * WAVE + lane;
                if (base == 0) raw[g] = raw_first[g];
                else raw[g] = idx < n_eff ? recs[idx] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < GR; g++) {
                const uint32_t idx = base + g * WAVE + lane;
                const bool unmapped = ((raw[g].x >> 30) & 1u) != 0;
                const uint32_t contig = raw[g].y & 0xFFFFu;
                const bool primary = idx == 0 || idx == j2;
                if (idx < n_eff && !unmapped) {                                 // an unmapped record has no alignment; unmapped primaries end the pair above
                    const uint32_t e = idx >= split ? 1u : 0u;
                    if (contig >= A) {
                        if (!primary) { if (e) bad1 = 1; else bad0 = 1; }
                    } else {
                        const Scored sc = score_counted(L, raw[g]);
                        if (e == 0) { be0 = min(be0, sc.edit); bl0 = fmax(bl0, sc.ln_prob); }
                        else { be1 = min(be1, sc.edit); bl1 = fmax(bl1, sc.ln_prob); }
                        if (sc.edit <= (e ? pass1 : pass0)) {                   // save (locs.rs:314)
                            if constexpr (KEEP) {
                                // 28 + 1 + 18 (three 16-bit counts) + 16 bits (the kernel is not launched on pairs of 65535 records or more)
                                const unsigned long long packed = static_cast<unsigned long long>(sc.start) | static_cast<unsigned long long>((raw[g].x >> 28) & 1u) << 28 |
                                    static_cast<unsigned long long>(sc.end - sc.start) << 29 | static_cast<unsigned long long>(idx) << 47;
                                const unsigned long long before = atomicExch(&kept[e * A + contig], packed);
                                if (before != NOT_KEPT) {
                                    if constexpr (TWO) {
                                        // the group's second saved alignment goes on the pair's list (which of the two stays in the slot is
                                        // all one: pass 3 orders them; their ln-probabilities are computed again there — kept_lp holds either)
                                        const uint32_t at = atomicAdd(ovf_n, 1u);
                                        if (at < LEAN_OVF) {
                                            ovf_val[at] = before;
                                            multi |= atomicExch(&ovf_at[e * A + contig], at) != LEAN_NONE;      // a third one: the general kernel's
                                        } else multi = true;
                                    } else multi = true;
                                }
                                kept_lp[e * A + contig] = sc.ln_prob;
                            } else multi |= atomicExch(&head[e * A + contig], idx) != 0xFFFFFFFFu;
                        }
                    }
                }
            }
        }
        be0 = wave_min_u32(be0); be1 = wave_min_u32(be1);
        bl0 = wave_max_f64(bl0); bl1 = wave_max_f64(bl1);
        bad0 = wave_sum_u32(bad0); bad1 = wave_sum_u32(bad1);
        __syncthreads();
        stamp(1);
        if (!regs_ok || __ballot(multi) != 0ull) {
            // not this kernel's pair: untouched, the general kernel takes it
            if (lane == 0) R.defer_list[atomicAdd(R.defer_count, 1u)] = static_cast<uint32_t>(p);
            continue;
        }

        const bool wm0 = look && be0 <= (L.strict_subset ? pass0 : thr0) && !bad0;
        const bool saved1 = (st1 & 3u) == 3u;
        const bool err = (st0 & 4u) || (look && bad0) || (wm0 && ((st1 & 4u) || (saved1 && bad1)));
        if (err && lane == 0) atomicMax(R.err_flag, static_cast<uint32_t>(LCTY_ERR_INVALID_DATA));
        const bool wm1 = !paired || (saved1 && be1 <= (L.strict_subset ? pass1 : thr1));
        bool accepted = wm0 && wm1 && !err;
        double weight = 1.0;
        if (accepted) {
            weight *= be0 <= good0 ? 1.0 : sqrt(static_cast<double>(good0) / static_cast<double>(be0));      // locs.rs:565
            if (paired) weight *= be1 <= good1 ? 1.0 : sqrt(static_cast<double>(good1) / static_cast<double>(be1));
        }
        uint8_t status = LCTY_READ_POORLY_MAPPED;
        uint32_t total_cnt = 0, uk0 = 0, uk1 = 0;
        double unmapped_prob = 0.0;
        uint64_t pa_base = 0;
        if (!accepted && lane == 0) R.recover_w[p] = -1.0;
        if (accepted) {
            pair_unique_kmers_regs(L.kset, L.kset_mask, L.undef_in_set, L.k, len0, len1, bw0, bw1, nm0, nm1, lane, &uk0, &uk1);
            stamp(2);
            const uint32_t paired_count = (uk0 + uk1) & 0xFFFFu;
            double kw = L.weight_interc + static_cast<double>(paired_count) * L.weight_mult;
            kw = kw < 0.0 ? 0.0 : (kw > 1.0 ? 1.0 : kw);
            weight *= kw;
            const uint32_t max_alns = weight >= L.min_weight ? LCTY_MAX_USED_ALNS : LCTY_MAX_UNUSED_ALNS;
            const double unm_ins_penalty = L.unmapped_penalty + L.insert_penalty;
            unmapped_prob = paired ? weight * (2.0 * L.unmapped_penalty + L.insert_penalty) : weight * L.unmapped_penalty;
            const AlnRef none{0.0, 0, 0, NONE32, false};
            auto candidates = [&](uint32_t c) -> Fast3 {
                if constexpr (KEEP) {
                    const unsigned long long k1 = kept[c], k2 = kept[A + c];
                    const bool has1 = k1 != NOT_KEPT, has2 = k2 != NOT_KEPT;
                    auto ref = [](unsigned long long v, double lp) {
                        const uint32_t start = static_cast<uint32_t>(v) & 0x0FFFFFFFu;
                        return AlnRef{lp, start, start + (static_cast<uint32_t>(v >> 29) & 0x3FFFFu), static_cast<uint32_t>(v >> 47), ((v >> 28) & 1ull) != 0};
                    };
                    return fast_from_refs(ins, has1 ? ref(k1, kept_lp[c] - bl0) : none, has2 ? ref(k2, kept_lp[A + c] - bl1) : none,
                                          has1, has2, unm_ins_penalty, paired);
                } else {
                    const uint32_t h1 = head[c], h2 = head[A + c];
                    const bool has1 = h1 != 0xFFFFFFFFu, has2 = h2 != 0xFFFFFFFFu;
                    const uint4 r1 = recs[has1 ? h1 : 0u], r2 = recs[has2 ? h2 : 0u];      // L2: read in pass 1 a moment ago
                    return fast_from_refs(ins, has1 ? ref_from_counted(L, r1, h1, bl0) : none, has2 ? ref_from_counted(L, r2, h2, bl1) : none,
                                          has1, has2, unm_ins_penalty, paired);
                }
            };
            // TWO: a (contig, read end) group may hold a second saved alignment (the pair's list); such a contig takes the general pairing
            // over at most 2 x 2 alignments in registers (SmallCtx), every other contig the three candidates of the fast path
            SmallCtx sc;
            sc.ins = ins; sc.unm_ins_penalty = unm_ins_penalty; sc.paired = paired; sc.k1 = sc.k2 = 0;
            sc.a10 = sc.a11 = sc.a20 = sc.a21 = none;
            auto packed_ref = [](unsigned long long v, double lp) {
                const uint32_t start = static_cast<uint32_t>(v) & 0x0FFFFFFFu;
                return AlnRef{lp, start, start + (static_cast<uint32_t>(v >> 29) & 0x3FFFFu), static_cast<uint32_t>(v >> 47), ((v >> 28) & 1ull) != 0};
            };
            // the saved alignments of one group in the order of gather_sort_dedupe: ln_prob descending, record index ascending, the later
            // member of a 128-bp bin dropped (locs.rs:321-342); with two of them their ln-probabilities come from the records again
            struct Group { AlnRef r0, r1; uint32_t k; };
            auto group_refs = [&](uint32_t key, double best) -> Group {
                Group g{none, none, 0u};
                if constexpr (KEEP) {
                    const unsigned long long m = kept[key];
                    if (m == NOT_KEPT) return g;
                    unsigned long long x = NOT_KEPT;
                    if constexpr (TWO) { const uint32_t at = ovf_at[key]; if (at != LEAN_NONE) x = ovf_val[at]; }
                    if (x == NOT_KEPT) { g.r0 = packed_ref(m, kept_lp[key] - best); g.k = 1u; return g; }
                    const uint32_t i0 = static_cast<uint32_t>(m >> 47), i1 = static_cast<uint32_t>(x >> 47);
                    const double l0 = score_counted(L, recs[i0]).ln_prob, l1 = score_counted(L, recs[i1]).ln_prob;
                    const bool m_first = l0 > l1 || (l0 == l1 && i0 < i1);
                    g.r0 = packed_ref(m_first ? m : x, (m_first ? l0 : l1) - best);
                    g.r1 = packed_ref(m_first ? x : m, (m_first ? l1 : l0) - best);
                    g.k = (g.r0.start >> 7) == (g.r1.start >> 7) ? 1u : 2u;
                }
                return g;
            };
            // the candidates of contig c: true = through `sc` (some group holds two), false = through the fast path's three
            auto build = [&](uint32_t c, Fast3& f) -> bool {
                if constexpr (TWO) {
                    const Group g1 = group_refs(c, bl0), g2 = group_refs(A + c, bl1);
                    sc.a10 = g1.r0; sc.a11 = g1.r1; sc.k1 = g1.k; sc.a20 = g2.r0; sc.a21 = g2.r1; sc.k2 = g2.k;
                    if (sc.k1 > 1 || sc.k2 > 1) return true;
                    f = fast_from_refs(ins, g1.r0, g2.r0, sc.k1 != 0, sc.k2 != 0, unm_ins_penalty, paired);
                    return false;
                } else { f = candidates(c); return false; }
            };
            // Where the entries go is decided BEFORE the contigs are looked at when the wavefront's share of the arena can hold the
            // most this pair could write (3 entries per contig that holds a saved alignment): the candidates are then built once,
            // counted and written in the same pass. Otherwise (no shares in a small launch, or a pair too big for one) the count
            // comes first and the candidates are built a second time for the writing.
            uint32_t ub = 0;
            for (uint32_t c0 = 0; c0 < A; c0 += WAVE) {
                const uint32_t c = c0 + lane;
                if constexpr (KEEP) ub += __popcll(__ballot(c < A && (kept[c] & kept[A + c]) != NOT_KEPT));
                else ub += __popcll(__ballot(c < A && (head[c] & head[A + c]) != 0xFFFFFFFFu));
            }
            ub *= 3;
            const bool at_once = !TWO && R.pa_chunk != 0 && static_cast<uint64_t>(ub) * 8 <= R.pa_chunk;
            bool room = true;
            if (at_once) {
                if (ub > pool_left) {
                    if (lane == 0) pool_at = atomicAdd(R.pa_count, static_cast<unsigned long long>(R.pa_chunk));
                    pool_at = __shfl(pool_at, 0);
                    pool_left = R.pa_chunk;
                }
                pa_base = pool_at;
                room = pa_base + ub <= R.pa_cap;
            }
            stamp(3);
            // ---------------- pass 3a ----------------
            bool inb = false;
            uint32_t run = 0;
            for (uint32_t c0 = 0; c0 < A; c0 += WAVE) {
                const uint32_t c = c0 + lane;
                ContigResult res{-INFINITY, 0};
                Fast3 f;
                f.has1 = f.has2 = false;
                if (c < A) {
                    const bool small = build(c, f);
                    const uint32_t clen = L.allele_len[c];
                    if (small) {
                        for (uint32_t i = 0; i < sc.k1; i++) { const AlnRef a = sc.get1(i); const uint32_t mid = (a.start + a.end) / 2; inb |= L.boundary <= mid && mid < clen - L.boundary; }
                        for (uint32_t j = 0; j < sc.k2; j++) { const AlnRef a = sc.get2(j); const uint32_t mid = (a.start + a.end) / 2; inb |= L.boundary <= mid && mid < clen - L.boundary; }
                        res = count_pairs(sc, max_alns, L.prob_diff);
                    } else {
                        if (f.has1) { const uint32_t mid = (f.a1.start + f.a1.end) / 2; inb |= L.boundary <= mid && mid < clen - L.boundary; }
                        if (f.has2) { const uint32_t mid = (f.a2.start + f.a2.end) / 2; inb |= L.boundary <= mid && mid < clen - L.boundary; }
                        if (f.has1 || f.has2) res = fast_count(f, max_alns, L.prob_diff);
                    }
                    cnt8[c] = static_cast<uint8_t>(res.cnt);
                    mrow[c] = res.cnt ? res.best * weight : unmapped_prob;
                    total_cnt += res.cnt;
                }
                if (at_once) {
                    uint32_t tot;
                    const uint32_t my_off = run + wave_excl_scan_u32(res.cnt, lane, &tot);
                    run += tot;
                    if (c < A) R.pa_idx[p * A + c] = my_off | (res.cnt << 24);
                    if (room && res.cnt) fast_emit(f, res.cnt, weight, c, R.pa + pa_base + my_off);
                }
            }
            stamp(4);
            const bool any_inb = __ballot(inb) != 0ull;
            total_cnt = wave_sum_u32(total_cnt);
            if (lane == 0) R.recover_w[p] = any_inb ? weight : -1.0;
            const bool edit_good = be0 <= thr0 && (!paired || be1 <= thr1);
            if (!any_inb) { status = LCTY_READ_OUT_OF_BOUNDS; accepted = false; }
            else if (!edit_good) { status = LCTY_READ_POORLY_MAPPED; accepted = false; }
            else status = weight >= L.min_weight ? LCTY_READ_GOOD : LCTY_READ_FEW_KMERS;
            if (accepted && at_once) {
                // what a pair that is not kept wrote stays behind the share's mark and is written over by the next pair
                if (!room && lane == 0) atomicMax(R.err_flag, static_cast<uint32_t>(LCTY_ERR_RUNTIME));
                pool_at += total_cnt; pool_left -= total_cnt;
            } else if (accepted) {
                if (R.pa_chunk == 0 || static_cast<uint64_t>(total_cnt) * 8 > R.pa_chunk) {
                    if (lane == 0) pa_base = atomicAdd(R.pa_count, static_cast<unsigned long long>(total_cnt));
                    pa_base = __shfl(pa_base, 0);
                } else {
                    if (total_cnt > pool_left) {
                        if (lane == 0) pool_at = atomicAdd(R.pa_count, static_cast<unsigned long long>(R.pa_chunk));
                        pool_at = __shfl(pool_at, 0);
                        pool_left = R.pa_chunk;
                    }
                    pa_base = pool_at;
                    pool_at += total_cnt; pool_left -= total_cnt;
                }
                if (lane == 0 && pa_base + total_cnt > R.pa_cap) atomicMax(R.err_flag, static_cast<uint32_t>(LCTY_ERR_RUNTIME));
                room = pa_base + total_cnt <= R.pa_cap;
                // ---------------- pass 3b: every lane wrote its own cnt8 entries: no barrier in between ----------------
                run = 0;
                for (uint32_t c0 = 0; c0 < A; c0 += WAVE) {
                    const uint32_t c = c0 + lane;
                    const uint32_t cnt = c < A ? cnt8[c] : 0u;
                    uint32_t tot;
                    const uint32_t my_off = run + wave_excl_scan_u32(cnt, lane, &tot);
                    run += tot;
                    if (c < A) R.pa_idx[p * A + c] = my_off | (cnt << 24);
                    if (room && cnt) {
                        Fast3 f2;
                        f2.has1 = f2.has2 = false;
                        if (build(c, f2)) emit_pairs(sc, cnt, weight, c, R.pa + pa_base + my_off);
                        else fast_emit(f2, cnt, weight, c, R.pa + pa_base + my_off);
                    }
                }
            }
        }
        if (status != LCTY_READ_GOOD)
            for (uint32_t c = lane; c < A; c += WAVE) mrow[c] = 0.0;
        if (lane == 0) {
            R.status[p] = status;
            R.weight[p] = accepted ? weight : 0.0;
            R.unmapped_prob[p] = accepted ? unmapped_prob : 0.0;
            R.pa_off[p] = accepted ? pa_base : 0ull;
            R.pa_cnt[p] = accepted ? total_cnt : 0u;
            R.uniq_kmers[2 * p] = accepted ? static_cast<uint16_t>(uk0) : 0;
            R.uniq_kmers[2 * p + 1] = accepted ? static_cast<uint16_t>(uk1) : 0;
        }
        __syncthreads();
        stamp(5);
        if constexpr (TIMED) tph[7] += 1;
    }
    if constexpr (TIMED) {
        if (lane == 0)
            for (int k = 0; k < 8; k++) atomicAdd(R.dbg + k, static_cast<unsigned long long>(tph[k]));
    }
}

__global__ __launch_bounds__(WAVE, 4) void score_counted_lean_kernel(const LocusView L, const ReadsView R) { score_lean_body<false, false>(L, R); }
__global__ __launch_bounds__(WAVE, 4) void score_counted_lean_keep_kernel(const LocusView L, const ReadsView R) { score_lean_body<false, true>(L, R); }
// the pairs the kernel above left, with a second saved alignment per (contig, read end) allowed (13 % of the pairs at config 2; what it
// leaves in turn — a group of three or more, or more than LEAN_OVF such groups — is the general kernel's)
__global__ __launch_bounds__(WAVE, 2) void score_counted_lean_two_kernel(const LocusView L, const ReadsView R) { score_lean_body<false, true, true>(L, R); }
#ifdef LCTY_DIAG     // the developer build: the same kernels with shader-clock stamps between their phases (knob "score_timing")
__global__ __launch_bounds__(WAVE, 4) void score_counted_lean_timed_kernel(const LocusView L, const ReadsView R) { score_lean_body<true, false>(L, R); }
__global__ __launch_bounds__(WAVE, 4) void score_counted_lean_keep_timed_kernel(const LocusView L, const ReadsView R) { score_lean_body<true, true>(L, R); }
#endif

__global__ __launch_bounds__(WAVE, 3) void score_reads_kernel(const LocusView L, const ReadsView R, const uint32_t max_recs) {
    score_reads_body<false, false>(L, R, max_recs);
}
// the same with explicit region weights (--reg-weights)
__global__ __launch_bounds__(WAVE, 3) void score_reads_explicit_kernel(const LocusView L, const ReadsView R, const uint32_t max_recs) {
    score_reads_body<true, false>(L, R, max_recs);
}
// the same two for read pairs whose saved alignments do not fit the LDS (thousands of alleles): parked in global memory
__global__ __launch_bounds__(WAVE, 3) void score_reads_big_kernel(const LocusView L, const ReadsView R, const uint32_t max_recs) {
    score_reads_body<false, true>(L, R, max_recs);
}
__global__ __launch_bounds__(WAVE, 3) void score_reads_big_explicit_kernel(const LocusView L, const ReadsView R, const uint32_t max_recs) {
    score_reads_body<true, true>(L, R, max_recs);
}

// the same four for batches of counted alignments (lcty_reads_append_counted): no CIGAR words to decode
__global__ __launch_bounds__(WAVE, 3) void score_counted_kernel(const LocusView L, const ReadsView R, const uint32_t max_recs) {
    score_reads_body<false, false, true>(L, R, max_recs);
}
__global__ __launch_bounds__(WAVE, 3) void score_counted_explicit_kernel(const LocusView L, const ReadsView R, const uint32_t max_recs) {
    score_reads_body<true, false, true>(L, R, max_recs);
}
__global__ __launch_bounds__(WAVE, 3) void score_counted_big_kernel(const LocusView L, const ReadsView R, const uint32_t max_recs) {
    score_reads_body<false, true, true>(L, R, max_recs);
}
__global__ __launch_bounds__(WAVE, 3) void score_counted_big_explicit_kernel(const LocusView L, const ReadsView R, const uint32_t max_recs) {
    score_reads_body<true, true, true>(L, R, max_recs);
}

static size_t score_lds_bytes(uint32_t max_recs, uint32_t A) {
    const size_t mr2 = (max_recs + 1) & ~1u;
    size_t b = static_cast<size_t>(max_recs) * sizeof(Rec16) + static_cast<size_t>(A) * 8 + 4;   // rec + head + alen + cursor
    b += 2 * mr2 * sizeof(uint16_t) + static_cast<size_t>(3) * A;                                // nxt + order + kk1/kk2/cnt8
    return (b + 15) & ~static_cast<size_t>(15);
}

// BIG variant: the parked alignments (global scratch per workgroup) and the per-allele tables (LDS)
static size_t score_park_bytes(uint32_t max_recs, uint32_t A) {
    const size_t mr2 = (max_recs + 1) & ~1u;
    return (static_cast<size_t>(max_recs) * sizeof(Rec16) + 2 * mr2 * sizeof(uint16_t) + static_cast<size_t>(3) * A + 255) & ~static_cast<size_t>(255);
}
static size_t score_table_bytes(uint32_t A) {
    return (static_cast<size_t>(A) * 4 + 4 + 15) & ~static_cast<size_t>(15);
}

void launch_score_reads(lcty_reads* reads) {
    lcty_ctx* ctx = reads->ctx;
    const LocusView L = reads->locus->view();
    ReadsView R = reads->view();
    const uint32_t max_recs = std::max<uint32_t>(reads->max_recs_per_pair, 1);
    size_t lds = score_lds_bytes(max_recs, L.n_alleles);
    const size_t lds_max = 160 * 1024;
    if (max_recs >= 65535) fail(LCTY_ERR_UNSUPPORTED, "more than 65534 records in one read pair");
    // the saved alignments of a pair live in LDS when they fit; beyond that (thousands of alleles, an alignment on each) in a
    // per-workgroup scratch in global memory, only the per-allele tables in LDS
    const bool big = lds > lds_max;
    if (big) {
        lds = score_table_bytes(L.n_alleles);
        if (lds > lds_max)
            fail(LCTY_ERR_UNSUPPORTED, "%u alleles need %zu B of LDS for the per-allele tables (> %zu): not supported by this build",
                 L.n_alleles, lds, lds_max);
    }
    const bool explicit_weights = L.ew_val != nullptr;
    auto kernel = reads->counted
        ? (big ? (explicit_weights ? score_counted_big_explicit_kernel : score_counted_big_kernel)
               : (explicit_weights ? score_counted_explicit_kernel : score_counted_kernel))
        : (big ? (explicit_weights ? score_reads_big_explicit_kernel : score_reads_big_kernel)
               : (explicit_weights ? score_reads_explicit_kernel : score_reads_kernel));
    if (lds > 48 * 1024)
        LCTY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    const uint32_t cus = static_cast<uint32_t>(ctx->props.multiProcessorCount);
    const uint32_t per_cu = static_cast<uint32_t>(std::max<size_t>(1, std::min<size_t>(12, lds_max / lds)));
    const uint64_t grid = std::max<uint64_t>(1, std::min<uint64_t>(R.n_pairs, static_cast<uint64_t>(cus) * per_cu));
    if (big) {
        const size_t stride = score_park_bytes(max_recs, L.n_alleles);
        reads->d_park.ensure(stride * grid);
        R.park = reads->d_park.p; R.park_stride = stride;
    }
    // chunk-wise reservation abandons up to one partly used chunk per wavefront and LAUNCH: the arena's slack (lcty_reads_create) covers
    // PA_MAX_GRID wavefronts of one launch, so larger grids (parts with more CUs) and the many launches of a streaming batch reserve per pair
    const bool pooled = reads->pa_pooled && !reads->streaming;
    R.pa_chunk = pooled && grid <= PA_MAX_GRID && R.n_pairs / grid >= PA_POOL_MIN_PAIRS ? PA_CHUNK : 0u;
    // the arena cursor goes back to where the pairs on the device start (0 unless a streaming batch has dropped chunks)
    if (reads->raw_first == 0) reads->d_pa_count.zero(ctx->stream);
    else reads->d_pa_count.upload(&reads->pa_at_raw_first, 1, ctx->stream);
    R.defer_list = nullptr; R.defer_count = nullptr; R.only_list = nullptr; R.only_count = nullptr;
    // Counted batches: the lean kernel first (at most one saved alignment per contig and read end: the rule), the general kernel
    // on the pairs it left. lcty_ctx_set_knob "score_lean" 0: the general kernel on everything.
    const bool lean = reads->counted && !big && !explicit_weights && L.k <= 31 && L.n_alleles <= 16000 && ctx->knob("score_lean", 1) != 0;
    if (lean) {
        reads->d_defer_list.ensure(std::max<uint64_t>(R.n_pairs, 1)); reads->d_defer_count.ensure(1);
        reads->d_defer_count.zero(ctx->stream);
        R.defer_list = reads->d_defer_list.p; R.defer_count = reads->d_defer_count.p;
    }
    ctx->timed(LCTY_K_SCORE, [&] {
        if (lean) {
            // sixteen wavefronts per CU either way: with the saved alignments' products kept in LDS (33 B per allele) while that many fit
            const bool keep = static_cast<size_t>(L.n_alleles) * 33 + 16 <= lds_max / 16 && ctx->knob("score_lean_keep", 1) != 0;
            const size_t lean_lds = (static_cast<size_t>(L.n_alleles) * (keep ? 33 : 9) + 15) & ~static_cast<size_t>(15);
            auto lean_kernel = keep ? score_counted_lean_keep_kernel : score_counted_lean_kernel;
#ifdef LCTY_DIAG
            const bool timing = ctx->diag_knob("score_timing", 0) != 0;
            if (timing) lean_kernel = keep ? score_counted_lean_keep_timed_kernel : score_counted_lean_timed_kernel;
#else
            constexpr bool timing = false;
#endif
            if (lean_lds > 48 * 1024)
                LCTY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lean_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(lean_lds)));
            const uint32_t lean_per_cu = static_cast<uint32_t>(std::max<size_t>(1, std::min<size_t>(16, lds_max / lean_lds)));
            const uint64_t lean_grid = std::max<uint64_t>(1, std::min<uint64_t>(R.n_pairs, static_cast<uint64_t>(cus) * lean_per_cu));
            ReadsView RL = R;
            RL.pa_chunk = pooled && lean_grid <= PA_MAX_GRID && R.n_pairs / lean_grid >= PA_POOL_MIN_PAIRS ? PA_CHUNK : 0u;
            if (timing) {
                // diagnostic: the timed build of the lean kernel; ticks (10 ns) per pair of a wavefront, phase by phase, on stderr
                reads->d_score_dbg.ensure(8); reads->d_score_dbg.zero(ctx->stream);
                RL.dbg = reads->d_score_dbg.p;
                hipLaunchKernelGGL(lean_kernel, dim3(static_cast<uint32_t>(lean_grid)), dim3(WAVE), lean_lds, ctx->stream, L, RL);
                unsigned long long t[8];
                reads->d_score_dbg.download(t, 8, ctx->stream);
                LCTY_HIP(hipStreamSynchronize(ctx->stream));
                const double n = static_cast<double>(std::max<unsigned long long>(t[7], 1));
                fprintf(stderr, "[lcty score] lean kernel, ticks per pair of a wavefront (%llu pairs, %llu wavefronts): records + thresholds %.0f, pass 1 %.0f, "
                        "k-mers %.0f, arena share %.0f, pass 3 %.0f, tail %.0f\n", t[7], static_cast<unsigned long long>(lean_grid),
                        t[0] / n, t[1] / n, t[2] / n, t[3] / n, t[4] / n, t[5] / n);
            } else hipLaunchKernelGGL(lean_kernel, dim3(static_cast<uint32_t>(lean_grid)), dim3(WAVE), lean_lds, ctx->stream, L, RL);
            R.only_list = R.defer_list; R.only_count = R.defer_count;
            R.pa_chunk = 0;                                                     // the few pairs left reserve their own entries
            if (keep && ctx->knob("score_lean_two", 1) != 0) {
                // the pairs the lean kernel left go through its two-slot form first (a second saved alignment per (contig, read end):
                // most of them); the general kernel takes what that leaves. The counts stay on the device.
                reads->d_defer_list2.ensure(std::max<uint64_t>(R.n_pairs, 1)); reads->d_defer_count2.ensure(1);
                reads->d_defer_count2.zero(ctx->stream);
                ReadsView R2 = R;
                R2.defer_list = reads->d_defer_list2.p; R2.defer_count = reads->d_defer_count2.p;
                const size_t two_lds = ((((static_cast<size_t>(L.n_alleles) * 33 + 7) & ~static_cast<size_t>(7)) + LEAN_OVF * 8 + 8 + static_cast<size_t>(L.n_alleles) * 8) + 15) & ~static_cast<size_t>(15);
                if (two_lds > 48 * 1024)
                    LCTY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(score_counted_lean_two_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(two_lds)));
                const uint32_t two_per_cu = static_cast<uint32_t>(std::max<size_t>(1, std::min<size_t>(8, lds_max / two_lds)));    // two wavefronts per SIMD: 2 x 2 alignments per contig in registers
                const uint64_t two_grid = std::max<uint64_t>(1, std::min<uint64_t>(R.n_pairs, static_cast<uint64_t>(cus) * two_per_cu));
                hipLaunchKernelGGL(score_counted_lean_two_kernel, dim3(static_cast<uint32_t>(two_grid)), dim3(WAVE), two_lds, ctx->stream, L, R2);
                R.only_list = R2.defer_list; R.only_count = R2.defer_count;
            }
        }
        hipLaunchKernelGGL(kernel, dim3(static_cast<uint32_t>(grid)), dim3(WAVE), lds, ctx->stream, L, R, max_recs);
    });
    LCTY_HIP(hipGetLastError());
}

}  // namespace lcty
