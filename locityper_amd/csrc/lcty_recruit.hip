// lcty_recruit.hip — minimizer read recruitment: which loci does a read (pair) belong to?
//   src/seq/recruit.rs (Params 43-105, MatchCount 236-367, TargetBuilder 663-756, Targets::recruit_* 848-996),
//   src/seq/kmers.rs:93-103 (fasthash mix), 243-331 (canonical minimizers), src/math/frac.rs:50-93.
//
// Targets are built on the host once (minimizers of the alleles, one entry per (minimizer, locus): direction bits, "rare");
// the device holds an open-addressing table minimizer -> run of entries. recruit_kernel: one lane per read pair. A lane walks
// a mate base by base (rolling forward / reverse k-mer, hash, ring of the last w hashes in LDS, window minimum with the rescan
// rule of the reference) and only WRITES its minimizers down (wavefront scratch, interleaved across the lanes); then all lanes
// probe the table with their j-th minimizer together — the lanes find their minimizers at different bases, and a probe inside
// the walk would run the table look-up (a dependent global load) once per base with one lane active. Matches are counted per
// locus in registers (up to 8 loci per read pair, linear search); then the fraction tests. Integer arithmetic throughout.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <unordered_map>

#include "lcty_objects.hpp"

namespace lcty {

constexpr uint64_t UNDEF64 = ~0ull;
constexpr uint32_t MAX_LOCI_PER_READ = 8;
constexpr uint32_t READ_LENGTH_THRESH = 500;       // recruit.rs:35

__host__ __device__ inline uint64_t fast_hash64(uint64_t x) {      // kmers.rs:93-103
    x = ~x;
    x ^= x >> 23;
    x *= 0x2127599bf4325c37ull;
    x ^= x >> 47;
    return x;
}

struct TableSlot { uint64_t key; uint32_t start, count; };      // key UNDEF64 = free (UNDEF is never a minimizer, kmers.rs:27-30)
// entry: locus | direction << 24 (bit 0 backward, bit 1 forward) | rare << 26

struct RecruitView {
    const TableSlot* table; uint64_t table_mask;
    const uint32_t* entries;
    uint32_t k, w, ring;                 // ring: power of two >= w
    uint32_t mf_num, mf_den;             // match_frac_short
    // reads
    uint64_t n_pairs;
    const uint32_t* mate_len; const uint64_t* mate_off; const uint32_t* bases2; const uint32_t* nmask;
    int paired;
    uint32_t max_out;
    uint32_t* out_cnt; uint32_t* out_loci;
    uint64_t* scratch; uint8_t* scratch_f; uint32_t scratch_cap;      // per workgroup [scratch_cap][64] minimizers of the mate being walked (hash, direction)
    // single reads of more than 256 bases: one wavefront per read
    const uint32_t* long_list; uint32_t n_long; uint32_t long_cap;      // reads of the list; minimizers a read can have
    uint64_t* long_h; uint8_t* long_f;             // per workgroup 2 x [long_cap]: minimizers in read order (second half: staging)
    double match_frac; uint32_t stretch_minims, stretch_score;
    uint32_t* err;
};

// per-lane match counts of up to MAX_LOCI_PER_READ loci; counts packed 4 x u16: [common-bw, common-fw, rare-bw, rare-fw] (recruit.rs:236-240)
struct Matches {
    uint32_t locus[MAX_LOCI_PER_READ];
    uint64_t first[MAX_LOCI_PER_READ], second[MAX_LOCI_PER_READ];
    uint32_t n;
};

// minimizers::<u64, _, CANONICAL> (kmers.rs:265-331) of one mate, as written there (ring of hashes, best position, rescan when the
// best leaves the window, the first_kmer / first_window bookkeeping around bases that are not ACGT); calls
// on_minimizer(hash, forward) for every minimizer; returns their number. The lanes of a wavefront rescan at different bases, so
// this form runs its rescan loop at almost every base: it is kept for the mates that contain such bases (and as the definition).
template <typename F>
__device__ inline uint32_t walk_minimizers(const RecruitView& V, const uint64_t* w64, const uint32_t* nm, uint32_t len, uint64_t* ring, F&& on_minimizer) {
    const uint32_t k = V.k, w = V.w, k_1 = k - 1, w_1 = w - 1, rmask = V.ring - 1;
    const uint64_t mask = (1ull << (2 * k)) - 1;
    const uint32_t rv_shift = 2 * k - 2;
    uint64_t fw_kmer = 0, rv_kmer = 0, fwd_bits = ~0ull;              // forward flag of ring slot j at bit j
    for (uint32_t j = 0; j < V.ring; j++) ring[j * 64] = UNDEF64;
    int64_t last_pos = -1;
    uint32_t best_pos = 0;
    uint64_t best_hash = UNDEF64;
    uint32_t first_kmer = k_1, first_window = k_1 + w_1, total = 0;
    uint64_t word = 0; uint32_t nword = 0;
    for (uint32_t i = 0; i < len; i++) {
        if ((i & 31u) == 0) { word = w64[i >> 5]; nword = nm[i >> 5]; }
        const uint64_t enc = (word >> ((i & 31u) * 2u)) & 3ull;
        const bool isn = (nword >> (i & 31u)) & 1u;
        uint64_t fw_enc = enc, rv_enc = 3ull - enc;
        if (isn) { first_kmer = i + k; fw_enc = 0; rv_enc = 0; }
        fw_kmer = ((fw_kmer << 2) | fw_enc) & mask;
        rv_kmer = (rv_kmer >> 2) | (rv_enc << rv_shift);
        const bool fwd = !(rv_kmer < fw_kmer);
        const uint64_t kmer = fwd ? fw_kmer : rv_kmer;
        const uint64_t h = i < first_kmer ? UNDEF64 : fast_hash64(kmer);
        ring[(i & rmask) * 64] = h;
        fwd_bits = (fwd_bits & ~(1ull << (i & rmask))) | (static_cast<uint64_t>(fwd) << (i & rmask));
        if (h < best_hash) { best_hash = h; best_pos = i; }
        if (i < first_window) continue;
        const uint32_t start = i - w_1;
        if (best_pos < start) {
            best_pos = start; best_hash = ring[(start & rmask) * 64];           // find_min (kmers.rs:243-258): leftmost minimum
            for (uint32_t j = start + 1; j <= i; j++) { const uint64_t v = ring[(j & rmask) * 64]; if (v < best_hash) { best_pos = j; best_hash = v; } }
            if (best_hash == UNDEF64) { first_window = first_window + w_1; continue; }
        }
        if (static_cast<int64_t>(best_pos) > last_pos) {
            last_pos = best_pos;
            total++;
            on_minimizer(best_hash, ((fwd_bits >> (best_pos & rmask)) & 1ull) != 0);
        }
    }
    return total;
}

// The same minimizers for a mate made of A, C, G, T only, without data-dependent branches: with every hash defined, the best
// position of the reference is simply the leftmost minimum of the last w k-mers, and a minimizer is reported whenever that
// position moves. Sliding minimum by blocks of w k-mers (van Herk / Gil-Werman): leftmost prefix minimum of the running block in
// registers, leftmost suffix minima of the block before it in LDS, written once per block by a backward pass — the same work
// at the same k-mer for all lanes. Returns false (nothing reported) if a hash happens to be UNDEF: the caller takes the form above.
template <typename F>
__device__ inline bool walk_minimizers_acgt(const RecruitView& V, const uint64_t* w64, uint32_t len, uint64_t* ring, uint64_t* suf_h, uint32_t* suf_p,
                                            uint32_t* total_out, F&& on_minimizer) {
    // ring / suf_h / suf_p: w slots each (not a power of two: the slot index is a counter that wraps), slot s of this lane at [s * 64]
    const uint32_t k = V.k, w = V.w, k_1 = k - 1, w_1 = w - 1;
    const uint64_t mask = (1ull << (2 * k)) - 1;
    const uint32_t rv_shift = 2 * k - 2;
    uint64_t fw_kmer = 0, rv_kmer = 0, fwd_bits = 0;                  // forward flag of k-mer t at bit t & 63 (only the last w matter, w < 64)
    uint64_t pre_h = UNDEF64; uint32_t pre_p = 0;                     // leftmost minimum of the running block
    uint32_t in_block = 0;                                            // k-mers of the running block so far = slot of the next k-mer
    int64_t last_pos = -1;
    uint32_t total = 0;
    bool undef_seen = false;
    uint64_t word = 0;
    for (uint32_t i = 0; i < len; i++) {
        if ((i & 31u) == 0) word = w64[i >> 5];
        const uint64_t enc = (word >> ((i & 31u) * 2u)) & 3ull;
        fw_kmer = ((fw_kmer << 2) | enc) & mask;
        rv_kmer = (rv_kmer >> 2) | ((3ull - enc) << rv_shift);
        if (i < k_1) continue;
        const uint32_t t = i - k_1;                                   // index of the k-mer that ends here; its slot is in_block (blocks start at slot 0)
        const bool fwd = !(rv_kmer < fw_kmer);
        const uint64_t h = fast_hash64(fwd ? fw_kmer : rv_kmer);
        undef_seen |= h == UNDEF64;
        const uint32_t slot = in_block;
        ring[slot * 64] = h;
        fwd_bits = (fwd_bits & ~(1ull << (t & 63u))) | (static_cast<uint64_t>(fwd) << (t & 63u));
        if (in_block == 0 || h < pre_h) { pre_h = h; pre_p = t; }
        in_block++;
        if (t >= w_1) {
            // window [t - w_1, t]: its part in the block before (suffix minimum from t - w_1 on: slot + 1 of that block) and the running block
            uint64_t best_h = pre_h; uint32_t best_p = pre_p;
            if (in_block < w) {
                const uint32_t sidx = (slot + 1) * 64;
                const uint64_t sh = suf_h[sidx];
                if (sh <= pre_h) { best_h = sh; best_p = suf_p[sidx]; }
            }
            if (static_cast<int64_t>(best_p) > last_pos) {
                last_pos = best_p;
                on_minimizer(best_h, ((fwd_bits >> (best_p & 63u)) & 1ull) != 0);
                total++;
            }
        }
        if (in_block == w) {                                          // block complete: its leftmost suffix minima, back to front
            uint64_t sh = UNDEF64; uint32_t sp = 0;
            for (uint32_t q = 0; q < w; q++) {
                const uint32_t s = w_1 - q;                               // slot of k-mer t - q
                const uint64_t v = ring[s * 64];
                if (q == 0 || v <= sh) { sh = v; sp = t - q; }
                suf_h[s * 64] = sh; suf_p[s * 64] = sp;
            }
            in_block = 0;
        }
    }
    *total_out = total;
    return !undef_seen;
}

__device__ __forceinline__ uint32_t c16(uint64_t packed, uint32_t i) { return static_cast<uint32_t>(packed >> (16 * i)) & 0xFFFFu; }
// (Wr + c) / (W(n - c) + c) with W = 3 (recruit.rs:282-325), all in u16 as upstream
__device__ __forceinline__ uint32_t fw_num(uint64_t a) { return (3u * c16(a, 3) + c16(a, 1)) & 0xFFFFu; }
__device__ __forceinline__ uint32_t bw_num(uint64_t a) { return (3u * c16(a, 2) + c16(a, 0)) & 0xFFFFu; }
__device__ __forceinline__ uint32_t fw_den(uint64_t a, uint32_t total) { return (3u * (total - c16(a, 1)) + c16(a, 1)) & 0xFFFFu; }
__device__ __forceinline__ uint32_t bw_den(uint64_t a, uint32_t total) { return (3u * (total - c16(a, 0)) + c16(a, 0)) & 0xFFFFu; }
__device__ __forceinline__ bool has_rare(uint64_t a) { return (a >> 32) != 0; }

__global__ __launch_bounds__(64) void recruit_kernel(const RecruitView V) {
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t lane = threadIdx.x;
    uint64_t* ring = reinterpret_cast<uint64_t*>(smem) + lane;                  // slot j of this lane at ring[j * 64]
    // the block-wise walk uses w slots of each of its three arrays; the walk as written uses V.ring slots of `ring` (same lane column)
    uint64_t* suf_h = ring + static_cast<size_t>(V.w) * 64;                     // suffix minima of the block before the running one
    uint32_t* suf_p = reinterpret_cast<uint32_t*>(reinterpret_cast<uint64_t*>(smem) + static_cast<size_t>(V.w) * 128) + lane;
    uint64_t* buf = V.scratch + static_cast<size_t>(blockIdx.x) * V.scratch_cap * 64 + lane;     // minimizer j of this lane at buf[j * 64]
    uint8_t* buf_f = V.scratch_f + static_cast<size_t>(blockIdx.x) * V.scratch_cap * 64 + lane;  // its direction
    for (uint64_t p0 = static_cast<uint64_t>(blockIdx.x) * 64; p0 < V.n_pairs; p0 += static_cast<uint64_t>(gridDim.x) * 64) {
        const uint64_t p = p0 + lane;
        const bool valid = p < V.n_pairs;
        const uint32_t len1 = valid ? V.mate_len[2 * p] : 0u, len2 = (valid && V.paired) ? V.mate_len[2 * p + 1] : 0u;
        bool bad = false;
        if (valid && !V.paired && len1 > 256u) bad = true;                           // taken by recruit_single_kernel, one wavefront per read
        Matches M; M.n = 0;
#pragma unroll
        for (uint32_t j = 0; j < MAX_LOCI_PER_READ; j++) { M.locus[j] = 0xFFFFFFFFu; M.first[j] = 0; M.second[j] = 0; }
        bool overflow = false;
        auto probe = [&](uint64_t minim, bool forward, bool second) {
            uint64_t slot = mix64(minim) & V.table_mask;
            TableSlot ts = V.table[slot];
            while (ts.key != UNDEF64 && ts.key != minim) { slot = (slot + 1) & V.table_mask; ts = V.table[slot]; }
            if (ts.key != minim) return;
            for (uint32_t q = 0; q < ts.count; q++) {
                const uint32_t e = V.entries[ts.start + q];
                const uint32_t locus = e & 0xFFFFFFu, direction = (e >> 24) & 3u, rare = (e >> 26) & 1u;
                // BaseMatchCount::inc (recruit.rs:250-256)
                const uint64_t inc = (static_cast<uint64_t>((direction & (1u + !forward)) != 0) << (32 * rare))
                                   | (static_cast<uint64_t>((direction & (1u + forward)) != 0) << (32 * rare + 16));
                bool found = false;
#pragma unroll
                for (uint32_t j = 0; j < MAX_LOCI_PER_READ; j++)
                    if (M.locus[j] == locus) { found = true; if (second) M.second[j] += inc; else M.first[j] += inc; }
                if (!found && !second) {                                         // get_or_insert; the second mate never inserts (912-915)
                    if (M.n == MAX_LOCI_PER_READ) overflow = true;
                    else {
#pragma unroll
                        for (uint32_t j = 0; j < MAX_LOCI_PER_READ; j++) if (j == M.n) { M.locus[j] = locus; M.first[j] = inc; }
                        M.n++;
                    }
                }
            }
        };
        uint32_t totals[2] = {0, 0};
        for (uint32_t mate = 0; mate < (V.paired ? 2u : 1u); mate++) {
            const uint32_t len = mate ? len2 : len1;
            // the second mate is only looked at when the first one matched something (recruit.rs:899)
            const bool walk = valid && !bad && (mate == 0 || M.n > 0);
            uint32_t total = 0;
            if (walk) {
                const uint64_t off = V.mate_off[2 * p + mate];
                const uint64_t* w64 = reinterpret_cast<const uint64_t*>(V.bases2) + (off >> 5);
                const uint32_t* nm = V.nmask + (off >> 5);
                uint32_t any_n = 0;
                for (uint32_t q = 0; q * 32 < len; q++) any_n |= nm[q] & (len - q * 32 >= 32 ? 0xFFFFFFFFu : ((1u << (len - q * 32)) - 1u));
                auto keep = [&](uint64_t h, bool fw) {
                    if (total < V.scratch_cap) { buf[static_cast<size_t>(total) * 64] = h; buf_f[static_cast<size_t>(total) * 64] = fw; }
                    total++;
                };
                bool done = false;
                if (any_n == 0) {
                    uint32_t n_fast = 0;
                    done = walk_minimizers_acgt(V, w64, len, ring, suf_h, suf_p, &n_fast, keep);
                    if (!done) total = 0;
                }
                if (!done) {
                    const uint32_t n_ref = walk_minimizers(V, w64, nm, len, ring, keep);
                    (void)n_ref;
                }
            }
            totals[mate] = total;
            uint32_t most = total;
            for (int o = 32; o > 0; o >>= 1) most = max(most, static_cast<uint32_t>(__shfl_xor(static_cast<int>(most), o)));
            for (uint32_t j = 0; j < most; j++) {
                if (j < total) probe(buf[static_cast<size_t>(j) * 64], buf_f[static_cast<size_t>(j) * 64] != 0, mate != 0);
            }
        }
        if (!valid || bad) continue;
        if (overflow) { atomicMax(V.err, static_cast<uint32_t>(LCTY_ERR_UNSUPPORTED)); V.out_cnt[p] = 0; continue; }
        const uint32_t total1 = totals[0], total2 = totals[1];
        uint32_t n_out = 0;
        uint32_t* out = V.out_loci + p * V.max_out;
#pragma unroll
        for (uint32_t j = 0; j < MAX_LOCI_PER_READ; j++) {
            if (j >= M.n) continue;
            const uint64_t a = M.first[j], b = M.second[j];
            bool take = false;
            if (V.paired) {
                if (has_rare(a) || has_rare(b)) {                                // recruit_read_pair (919-927), better_pair_fraction (351-367)
                    uint32_t n1, d1, n2, d2;
                    if (((fw_num(a) + bw_num(b)) & 0xFFFFu) >= ((bw_num(a) + fw_num(b)) & 0xFFFFu)) { n1 = fw_num(a); d1 = fw_den(a, total1); n2 = bw_num(b); d2 = bw_den(b, total2); }
                    else { n1 = bw_num(a); d1 = bw_den(a, total1); n2 = fw_num(b); d2 = fw_den(b, total2); }
                    take = n1 * V.mf_den >= V.mf_num * d1 && n2 * V.mf_den >= V.mf_num * d2;      // Fraction<u16>::partial_cmp (frac.rs:87-93)
                }
            } else if (has_rare(a)) {                                            // recruit_short_read (871-877), better_fraction (340-348)
                uint32_t n1, d1;
                if (fw_num(a) >= bw_num(a)) { n1 = fw_num(a); d1 = fw_den(a, total1); } else { n1 = bw_num(a); d1 = bw_den(a, total1); }
                take = n1 * V.mf_den >= V.mf_num * d1;
            }
            if (take) { if (n_out < V.max_out) out[n_out] = M.locus[j]; n_out++; }
        }
        if (n_out > V.max_out) atomicMax(V.err, static_cast<uint32_t>(LCTY_ERR_INVALID_INPUT));
        V.out_cnt[p] = min(n_out, V.max_out);
    }
}

// ---- single reads beyond the lane kernel's reach (recruit_short_read up to 500 bases, recruit_long_read beyond): one wavefront
// per read. The windows of the read are dealt to the lanes in 64 contiguous runs; a lane walks its run (plus the w k-mers before
// it, for the window that precedes its first one) with the block-wise sliding minimum and keeps the minimizers of its windows;
// the runs are then packed into one list in read order. Counting goes over the list 64 entries at a time into an LDS table of
// loci; has_matching_stretch — a running sum clamped at zero — is evaluated per lane as a function of the sum it starts with
// (s -> max(a, s + b), closed under composition) and the 64 pieces are chained.
constexpr uint32_t LONG_LOCI = 16;
struct LongTable { uint32_t locus[LONG_LOCI]; uint32_t cnt[LONG_LOCI][4]; uint32_t n; uint32_t total; uint32_t lane_cnt[64]; };

template <typename F>
__device__ inline void walk_run_acgt(const RecruitView& V, const uint64_t* w64, uint32_t t_own0, uint32_t t_own1, uint64_t* ring, uint64_t* suf_h,
                                     uint32_t* suf_p, bool* undef_seen, F&& on_minimizer) {
    // windows ending at k-mers [t_own0, t_own1) are this lane's; the window before the first one only fixes the last reported position
    const uint32_t k = V.k, w = V.w, k_1 = k - 1, w_1 = w - 1, rmask = V.ring - 1;
    const uint64_t mask = (1ull << (2 * k)) - 1;
    const uint32_t rv_shift = 2 * k - 2;
    const uint32_t ts = t_own0 >= w_1 + 1 ? t_own0 - 1 - w_1 : 0u;            // first k-mer needed
    uint64_t fw_kmer = 0, rv_kmer = 0, fwd_bits = 0;
    uint64_t pre_h = UNDEF64; uint32_t pre_p = 0, in_block = 0;
    int64_t last_pos = -1;
    uint64_t word = w64[ts >> 5];
    for (uint32_t i = ts; i < t_own1 + k_1; i++) {
        if ((i & 31u) == 0) word = w64[i >> 5];
        const uint64_t enc = (word >> ((i & 31u) * 2u)) & 3ull;
        fw_kmer = ((fw_kmer << 2) | enc) & mask;
        rv_kmer = (rv_kmer >> 2) | ((3ull - enc) << rv_shift);
        if (i < ts + k_1) continue;
        const uint32_t t = i - k_1;
        const bool fwd = !(rv_kmer < fw_kmer);
        const uint64_t h = fast_hash64(fwd ? fw_kmer : rv_kmer);
        *undef_seen |= h == UNDEF64;
        ring[(t & rmask) * 64] = h;
        fwd_bits = (fwd_bits & ~(1ull << (t & 63u))) | (static_cast<uint64_t>(fwd) << (t & 63u));
        if (in_block == 0 || h < pre_h) { pre_h = h; pre_p = t; }
        in_block++;
        if (t - ts >= w_1) {
            uint64_t best_h = pre_h; uint32_t best_p = pre_p;
            if (in_block < w) {
                const uint32_t sidx = ((t - w_1) & rmask) * 64;
                const uint64_t sh = suf_h[sidx];
                if (sh <= pre_h) { best_h = sh; best_p = suf_p[sidx]; }
            }
            if (static_cast<int64_t>(best_p) > last_pos) {
                last_pos = best_p;
                if (t >= t_own0) on_minimizer(best_h, ((fwd_bits >> (best_p & 63u)) & 1ull) != 0);
            }
        }
        if (in_block == w) {
            uint64_t sh = UNDEF64; uint32_t sp = 0;
            for (uint32_t q = 0; q < w; q++) {
                const uint32_t j = t - q;
                const uint64_t v = ring[(j & rmask) * 64];
                if (q == 0 || v <= sh) { sh = v; sp = j; }
                suf_h[(j & rmask) * 64] = sh; suf_p[(j & rmask) * 64] = sp;
            }
            in_block = 0;
        }
    }
}

__global__ __launch_bounds__(64) void recruit_single_kernel(const RecruitView V) {
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ LongTable T;
    const uint32_t lane = threadIdx.x;
    uint64_t* ring = reinterpret_cast<uint64_t*>(smem) + lane;
    uint64_t* suf_h = ring + static_cast<size_t>(V.ring) * 64;
    uint32_t* suf_p = reinterpret_cast<uint32_t*>(reinterpret_cast<uint64_t*>(smem) + static_cast<size_t>(V.ring) * 128) + lane;
    uint64_t* list_h = V.long_h + static_cast<size_t>(blockIdx.x) * 2 * V.long_cap;
    uint8_t* list_f = V.long_f + static_cast<size_t>(blockIdx.x) * 2 * V.long_cap;
    uint64_t* stage_h = list_h + V.long_cap; uint8_t* stage_f = list_f + V.long_cap;
    const uint32_t k = V.k, w_1 = V.w - 1;
    for (uint32_t r = blockIdx.x; r < V.n_long; r += gridDim.x) {
        const uint64_t p = V.long_list[r];
        const uint32_t len = V.mate_len[2 * p];
        const uint64_t off = V.mate_off[2 * p];
        const uint64_t* w64 = reinterpret_cast<const uint64_t*>(V.bases2) + (off >> 5);
        const uint32_t* nm = V.nmask + (off >> 5);
        // ---- the read's minimizers, in order ----
        uint32_t any_n = 0;
        for (uint32_t q = lane; q * 32 < len; q += 64) any_n |= nm[q] & (len - q * 32 >= 32 ? 0xFFFFFFFFu : ((1u << (len - q * 32)) - 1u));
        bool slow = __ballot(any_n != 0) != 0ull;
        uint32_t total = 0;
        const uint32_t n_kmers = len >= k ? len - k + 1 : 0u;
        if (!slow) {
            const uint32_t n_win = n_kmers > w_1 ? n_kmers - w_1 : 0u;
            const uint32_t per = (n_win + 63) / 64;
            const uint32_t t0 = w_1 + min(lane * per, n_win), t1 = w_1 + min((lane + 1) * per, n_win);
            uint32_t mine = 0;
            bool undef_seen = false;
            if (t1 > t0)
                walk_run_acgt(V, w64, t0, t1, ring, suf_h, suf_p, &undef_seen, [&](uint64_t h, bool fw) {
                    stage_h[lane * per + mine] = h; stage_f[lane * per + mine] = fw; mine++;
                });
            if (__ballot(undef_seen) != 0ull) slow = true;                       // a hash equal to UNDEF: by the book
            else {
                if (lane == 0) T.n = 0;
                T.lane_cnt[lane] = mine;
                __syncthreads();
                uint32_t before = 0;
                for (uint32_t q = 0; q < lane; q++) before += T.lane_cnt[q];
                for (uint32_t j = 0; j < mine; j++) { list_h[before + j] = stage_h[lane * per + j]; list_f[before + j] = stage_f[lane * per + j]; }
                total = before + mine;
                total = __shfl(static_cast<int>(total), 63);
            }
        }
        if (slow) {
            if (lane == 0) {
                uint32_t n = 0;
                walk_minimizers(V, w64, nm, len, ring, [&](uint64_t h, bool fw) { list_h[n] = h; list_f[n] = fw; n++; });
                T.total = n;
            }
            __syncthreads();
            total = T.total;
        }
        // ---- matches per locus (BaseMatchCount<u32>, recruit.rs:236-262) ----
        if (lane < LONG_LOCI) { T.locus[lane] = 0xFFFFFFFFu; T.cnt[lane][0] = T.cnt[lane][1] = T.cnt[lane][2] = T.cnt[lane][3] = 0; }
        __syncthreads();
        __threadfence_block();
        bool overflow = false;
        for (uint32_t j = lane; j < total; j += 64) {
            const uint64_t minim = list_h[j];
            const bool forward = list_f[j] != 0;
            uint64_t slot = mix64(minim) & V.table_mask;
            TableSlot ts = V.table[slot];
            while (ts.key != UNDEF64 && ts.key != minim) { slot = (slot + 1) & V.table_mask; ts = V.table[slot]; }
            if (ts.key != minim) continue;
            for (uint32_t q = 0; q < ts.count; q++) {
                const uint32_t e = V.entries[ts.start + q];
                const uint32_t locus = e & 0xFFFFFFu, direction = (e >> 24) & 3u, rare = (e >> 26) & 1u;
                uint32_t at = LONG_LOCI;
                for (uint32_t u = 0; u < LONG_LOCI; u++) {
                    const uint32_t old = atomicCAS(&T.locus[u], 0xFFFFFFFFu, locus);
                    if (old == 0xFFFFFFFFu || old == locus) { at = u; break; }
                }
                if (at == LONG_LOCI) { overflow = true; continue; }
                atomicAdd(&T.cnt[at][rare * 2], static_cast<uint32_t>((direction & (1u + !forward)) != 0));
                atomicAdd(&T.cnt[at][rare * 2 + 1], static_cast<uint32_t>((direction & (1u + forward)) != 0));
            }
        }
        __syncthreads();
        if (__ballot(overflow) != 0ull) { if (lane == 0) { atomicMax(V.err, static_cast<uint32_t>(LCTY_ERR_UNSUPPORTED)); V.out_cnt[p] = 0; } continue; }
        // ---- decisions ----
        uint32_t n_out = 0;
        for (uint32_t u = 0; u < LONG_LOCI; u++) {
            const uint32_t locus = T.locus[u];
            if (locus == 0xFFFFFFFFu) break;
            const uint32_t c0 = T.cnt[u][0], c1 = T.cnt[u][1], c2 = T.cnt[u][2], c3 = T.cnt[u][3];      // common-bw, common-fw, rare-bw, rare-fw
            bool take = false;
            if (len <= READ_LENGTH_THRESH) {                                     // recruit_short_read (871-877) in u16 arithmetic
                if (c2 || c3) {
                    const uint32_t fn = (3u * c3 + c1) & 0xFFFFu, bn = (3u * c2 + c0) & 0xFFFFu;
                    uint32_t n1, d1;
                    if (fn >= bn) { n1 = fn; d1 = (3u * (total - c1) + c1) & 0xFFFFu; } else { n1 = bn; d1 = (3u * (total - c0) + c0) & 0xFFFFu; }
                    take = n1 * V.mf_den >= V.mf_num * d1;
                }
            } else {                                                             // recruit_long_read (985-994)
                uint32_t num, den;
                if (c3 >= c2) { num = c3; den = total - c1; } else { num = c2; den = total - c0; }                 // rare_fraction (271-279)
                const uint32_t thr = max(1u, static_cast<uint32_t>(ceil(static_cast<double>(min(V.stretch_minims, den)) * V.match_frac)));
                take = num >= thr;
                if (take && den >= V.stretch_minims) {
                    // has_matching_stretch (938-961): s_fw / s_bw as functions of the sum the lane's piece starts with
                    const int64_t NEG = -(1ll << 40);
                    int64_t A[2] = {NEG, NEG}, B[2] = {0, 0}, Am[2] = {NEG, NEG}, Bm[2] = {NEG, NEG};
                    const uint32_t per = (total + 63) / 64;
                    for (uint32_t j = lane * per; j < min(total, (lane + 1) * per); j++) {
                        const uint64_t minim = list_h[j];
                        const bool forward = list_f[j] != 0;
                        int64_t add[2] = {0, 0};
                        uint64_t slot = mix64(minim) & V.table_mask;
                        TableSlot ts = V.table[slot];
                        while (ts.key != UNDEF64 && ts.key != minim) { slot = (slot + 1) & V.table_mask; ts = V.table[slot]; }
                        if (ts.key == minim)
                            for (uint32_t q = 0; q < ts.count; q++) {
                                const uint32_t e = V.entries[ts.start + q];
                                if ((e & 0xFFFFFFu) != locus) continue;
                                const uint32_t direction = (e >> 24) & 3u, x = 1u + ((e >> 26) & 1u) * 3u;           // SUBSUM_PENALTY + rare * SUBSUM_BONUS
                                add[0] = (direction & (1u + forward)) ? x : 0;
                                add[1] = (direction & (1u + !forward)) ? x : 0;
                            }
                        for (int d = 0; d < 2; d++) {                              // s = max(0, s + add - 1)
                            A[d] = max(static_cast<int64_t>(0), A[d] + add[d] - 1); B[d] += add[d] - 1;
                            Am[d] = max(Am[d], A[d]); Bm[d] = max(Bm[d], B[d]);
                        }
                    }
                    int64_t s[2] = {0, 0};
                    bool hit = false;
                    for (int l = 0; l < 64; l++) {
                        for (int d = 0; d < 2; d++) {
                            const int64_t a = __shfl(static_cast<long long>(A[d]), l), b = __shfl(static_cast<long long>(B[d]), l);
                            const int64_t am = __shfl(static_cast<long long>(Am[d]), l), bm = __shfl(static_cast<long long>(Bm[d]), l);
                            hit |= max(am, s[d] + bm) >= static_cast<int64_t>(V.stretch_score);
                            s[d] = max(a, s[d] + b);
                        }
                    }
                    take = hit;
                }
            }
            if (take) { if (lane == 0 && n_out < V.max_out) V.out_loci[p * V.max_out + n_out] = locus; n_out++; }
        }
        if (lane == 0) {
            if (n_out > V.max_out) atomicMax(V.err, static_cast<uint32_t>(LCTY_ERR_INVALID_INPUT));
            V.out_cnt[p] = min(n_out, V.max_out);
        }
        __syncthreads();
    }
}

}  // namespace lcty

using namespace lcty;

struct lcty_targets {
    lcty_ctx* ctx = nullptr;
    lcty_recruit_params prm{};
    uint16_t mf_num = 0, mf_den = 1;
    uint32_t n_loci = 0;
    bool finalized = false;
    struct Entry { uint32_t locus; uint8_t direction, rare; };
    std::unordered_map<uint64_t, std::vector<Entry>> minim_to_loci;
    DevBuf<TableSlot> d_table; DevBuf<uint32_t> d_entries, d_err;
    uint64_t table_mask = 0;
};

namespace {

// minimizers::<u64, _, CANONICAL> on the host (ASCII input), with positions: TargetBuilder::add needs them
void host_minimizers(const uint8_t* seq, size_t n, uint32_t k, uint32_t w, std::vector<uint32_t>& pos, std::vector<uint64_t>& hs, std::vector<uint8_t>& fwv) {
    pos.clear(); hs.clear(); fwv.clear();
    const uint64_t mask = (1ull << (2 * k)) - 1;
    const uint32_t rv_shift = 2 * k - 2, k_1 = k - 1, w_1 = w - 1;
    uint64_t fw_kmer = 0, rv_kmer = 0, hashes[64]; uint8_t forward[64];
    for (int i = 0; i < 64; i++) { hashes[i] = UNDEF64; forward[i] = 1; }
    int64_t last_pos = -1; uint32_t best_pos = 0; uint64_t best_hash = UNDEF64;
    uint32_t first_kmer = k_1, first_window = k_1 + w_1;
    for (size_t ii = 0; ii < n; ii++) {
        const uint32_t i = static_cast<uint32_t>(ii);
        uint64_t fe = 0, re = 0;
        switch (seq[ii]) {
            case 'A': fe = 0; re = 3; break;
            case 'C': fe = 1; re = 2; break;
            case 'G': fe = 2; re = 1; break;
            case 'T': fe = 3; re = 0; break;
            default: first_kmer = i + k;
        }
        fw_kmer = ((fw_kmer << 2) | fe) & mask;
        rv_kmer = (rv_kmer >> 2) | (re << rv_shift);
        const bool fwd = !(rv_kmer < fw_kmer);
        const uint64_t h = i < first_kmer ? UNDEF64 : fast_hash64(fwd ? fw_kmer : rv_kmer);
        hashes[i & 63] = h; forward[i & 63] = fwd;
        if (h < best_hash) { best_hash = h; best_pos = i; }
        if (i < first_window) continue;
        const uint32_t start = i - w_1;
        if (best_pos < start) {
            best_pos = start; best_hash = hashes[start & 63];
            for (uint32_t j = start + 1; j <= i; j++) if (hashes[j & 63] < best_hash) { best_pos = j; best_hash = hashes[j & 63]; }
            if (best_hash == UNDEF64) { first_window = first_window + w_1; continue; }
        }
        if (static_cast<int64_t>(best_pos) > last_pos) {
            last_pos = best_pos;
            pos.push_back(best_pos - k_1); hs.push_back(best_hash); fwv.push_back(forward[best_pos & 63]);
        }
    }
}

// Fraction::<u16>::approximate (frac.rs:50-76)
void approximate_u16(double x, uint16_t* num, uint16_t* den) {
    uint64_t a2 = 1, a1 = static_cast<uint64_t>(std::floor(x)), b2 = 0, b1 = 1;
    double xk = x;
    for (int it = 0; it < 20; it++) {
        const double numer = xk - std::floor(xk);
        if (numer <= 2.220446049250313e-16) break;
        xk = 1.0 / numer;
        const double fl = std::floor(xk);
        if (!(fl >= 0.0 && fl <= 65535.0)) break;
        const uint64_t f = static_cast<uint64_t>(fl);
        if (f * a1 > 65535 || f * a1 + a2 > 65535 || f * b1 > 65535 || f * b1 + b2 > 65535) break;
        const uint64_t a0 = f * a1 + a2, b0 = f * b1 + b2;
        a2 = a1; a1 = a0; b2 = b1; b1 = b0;
        if (std::fabs(static_cast<double>(a1) / static_cast<double>(b1) - x) <= 2.220446049250313e-16) break;
    }
    *num = static_cast<uint16_t>(a1); *den = static_cast<uint16_t>(b1);
}

}  // namespace

extern "C" {

int32_t lcty_recruit_params_default(lcty_recruit_params* p, int32_t technology, int32_t is_paired) {
    return guarded([&] {
        if (!p) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (technology != LCTY_TECH_ILLUMINA && is_paired) fail(LCTY_ERR_INVALID_INPUT, "Paired-end long reads are not supported");   // bg/mod.rs:250
        memset(p, 0, sizeof(*p));
        p->minimizer_k = 15; p->minimizer_w = 10;                               // recruit::DEFAULT_MINIM_KW
        p->match_length = 2000; p->thresh_kmer_count = 50;                      // recruit.rs:31, genotype.rs:139
        p->match_frac = technology == LCTY_TECH_ILLUMINA ? (is_paired ? 0.5 : 0.7) : 0.5;    // bg/mod.rs:245-252
    });
}

int32_t lcty_targets_create(lcty_ctx* ctx, const lcty_recruit_params* prm, lcty_targets** out) {
    return guarded([&] {
        if (!ctx || !prm || !out) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        // Params::new (recruit.rs:65-105)
        if (!(prm->minimizer_k > 0 && prm->minimizer_k <= 31)) fail(LCTY_ERR_INVALID_INPUT, "Minimizer kmer-size must be within [1, 31]");
        if (!(prm->minimizer_w > 1 && prm->minimizer_w <= 64)) fail(LCTY_ERR_INVALID_INPUT, "Minimizer window-size must be within [2, 64]");
        if (prm->minimizer_w == 64) fail(LCTY_ERR_UNSUPPORTED, "minimizer window of 64 k-mers");       // kmers.rs:268 asserts w < 64 as well
        if (!(prm->match_frac >= 0.25 && prm->match_frac <= 1.0)) fail(LCTY_ERR_INVALID_INPUT, "Minimizer match fraction (%g) must be in [0.25000, 1]", prm->match_frac);
        if (!(prm->match_length >= 200 && prm->match_length <= 100000)) fail(LCTY_ERR_INVALID_INPUT, "Matching stretch length (%u) should be between 200 and 100,000", prm->match_length);
        if (prm->thresh_kmer_count == 0) fail(LCTY_ERR_INVALID_INPUT, "k-mer threshold must be positive");
        auto t = std::make_unique<lcty_targets>();
        t->ctx = ctx; t->prm = *prm;
        approximate_u16(prm->match_frac, &t->mf_num, &t->mf_den);
        *out = t.release();
    });
}

void lcty_targets_destroy(lcty_targets* t) { delete t; }

int32_t lcty_targets_add_locus(lcty_targets* t, uint32_t n_alleles, const uint8_t* seqs, const uint64_t* seq_off, const uint16_t* counts,
                               const uint64_t* cnt_off, uint32_t base_k, uint32_t* locus_ix) {
    return guarded([&] {
        if (!t || !seqs || !seq_off || !counts || !cnt_off) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (t->finalized) fail(LCTY_ERR_INVALID_INPUT, "targets are finalized");
        if (t->n_loci >= (1u << 24) - 1) fail(LCTY_ERR_UNSUPPORTED, "too many loci");
        const uint32_t locus = t->n_loci, mk = t->prm.minimizer_k;
        const size_t shift = mk <= base_k ? (base_k - mk) / 2 : mk - base_k;
        std::vector<uint32_t> pos; std::vector<uint64_t> hs; std::vector<uint8_t> fw;
        for (uint32_t a = 0; a < n_alleles; a++) {
            const size_t len = seq_off[a + 1] - seq_off[a], n_counts = cnt_off[a + 1] - cnt_off[a];
            if ((len + 1 > base_k ? len + 1 - base_k : 0) != n_counts) fail(LCTY_ERR_INVALID_DATA, "Sequence and k-mer lengths do not match");   // recruit.rs:703
            const uint16_t* cnt = counts + cnt_off[a];
            host_minimizers(seqs + seq_off[a], len, mk, t->prm.minimizer_w, pos, hs, fw);
            for (size_t i = 0; i < hs.size(); i++) {
                const size_t p = pos[i];
                bool rare;
                if (mk <= base_k) rare = cnt[std::min<size_t>(p > shift ? p - shift : 0, n_counts - 1)] < t->prm.thresh_kmer_count;
                else rare = cnt[p] < t->prm.thresh_kmer_count && cnt[p + shift] < t->prm.thresh_kmer_count;
                auto& v = t->minim_to_loci[hs[i]];
                if (!v.empty() && v.back().locus == locus) { v.back().direction |= static_cast<uint8_t>(1 + fw[i]); v.back().rare &= static_cast<uint8_t>(rare); }
                else v.push_back(lcty_targets::Entry{locus, static_cast<uint8_t>(1 + fw[i]), static_cast<uint8_t>(rare)});
            }
        }
        t->n_loci++;
        if (locus_ix) *locus_ix = locus;
    });
}

int32_t lcty_targets_finalize(lcty_targets* t, uint64_t* n_minimizers) {
    return guarded([&] {
        if (!t) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (t->minim_to_loci.empty()) fail(LCTY_ERR_RUNTIME, "No minimizers for recruitment");       // recruit.rs:748
        uint64_t cap = 64;
        while (cap < 2 * t->minim_to_loci.size() + 2) cap <<= 1;
        std::vector<TableSlot> table(cap, TableSlot{UNDEF64, 0, 0});
        std::vector<uint32_t> entries;
        for (const auto& kv : t->minim_to_loci) {
            uint64_t slot = mix64(kv.first) & (cap - 1);
            while (table[slot].key != UNDEF64) slot = (slot + 1) & (cap - 1);
            table[slot] = TableSlot{kv.first, static_cast<uint32_t>(entries.size()), static_cast<uint32_t>(kv.second.size())};
            for (const auto& e : kv.second) entries.push_back(e.locus | (static_cast<uint32_t>(e.direction) << 24) | (static_cast<uint32_t>(e.rare) << 26));
        }
        lcty_ctx* ctx = t->ctx;
        ctx->activate();
        hipStream_t s = ctx->stream;
        t->d_table.alloc(cap); t->d_table.upload(table.data(), cap, s);
        t->d_entries.alloc(std::max<size_t>(entries.size(), 1)); t->d_entries.upload(entries.data(), entries.size(), s);
        t->d_err.alloc(1);
        LCTY_HIP(hipStreamSynchronize(s));
        t->table_mask = cap - 1;
        t->finalized = true;
        if (n_minimizers) *n_minimizers = t->minim_to_loci.size();
    });
}

int32_t lcty_recruit(lcty_targets* t, const lcty_reads_host* h, int32_t paired, uint32_t max_out, uint32_t* out_cnt, uint32_t* out_loci) {
    return guarded([&] {
        if (!t || !h || !out_cnt || !out_loci || max_out == 0) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (!t->finalized) fail(LCTY_ERR_INVALID_INPUT, "lcty_targets_finalize has not been called");
        const uint64_t n = h->n_pairs;
        if (n == 0) return;
        if (!h->mate_len || !h->mate_off || !h->bases2 || !h->nmask) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        lcty_ctx* ctx = t->ctx;
        ctx->activate();
        hipStream_t s = ctx->stream;
        const uint64_t nb = h->mate_off[2 * n];
        for (uint64_t m = 0; m < 2 * n; m++) {
            if (h->mate_off[m] % 32) fail(LCTY_ERR_INVALID_INPUT, "mate offsets must be multiples of 32 bases");
            if (h->mate_off[m + 1] < h->mate_off[m] + h->mate_len[m]) fail(LCTY_ERR_INVALID_INPUT, "mate offsets overlap");
        }
        DevBuf<uint32_t> d_len, d_bases, d_nm, d_cnt, d_loci; DevBuf<uint64_t> d_off;
        d_len.alloc(2 * n); d_off.alloc(2 * n + 1); d_bases.alloc(std::max<uint64_t>(nb / 16, 2)); d_nm.alloc(std::max<uint64_t>(nb / 32, 1));
        d_cnt.alloc(n); d_loci.alloc(n * max_out);
        d_len.upload(h->mate_len, 2 * n, s); d_off.upload(h->mate_off, 2 * n + 1, s);
        d_bases.upload(h->bases2, nb / 16, s); d_nm.upload(h->nmask, nb / 32, s);
        t->d_err.zero(s);
        RecruitView V{};
        V.table = t->d_table.p; V.table_mask = t->table_mask; V.entries = t->d_entries.p;
        V.k = t->prm.minimizer_k; V.w = t->prm.minimizer_w;
        V.ring = 2; while (V.ring < V.w) V.ring <<= 1;
        V.mf_num = t->mf_num; V.mf_den = t->mf_den;
        V.n_pairs = n; V.mate_len = d_len.p; V.mate_off = d_off.p; V.bases2 = d_bases.p; V.nmask = d_nm.p;
        V.paired = paired != 0; V.max_out = max_out; V.out_cnt = d_cnt.p; V.out_loci = d_loci.p; V.err = t->d_err.p;
        const size_t lds_single = static_cast<size_t>(V.ring) * 64 * (8 + 8 + 4);
        const size_t lds = std::max<size_t>(static_cast<size_t>(V.ring) * 8, static_cast<size_t>(V.w) * (8 + 8 + 4)) * 64;   // pair kernel: w slots per array
        // single reads of more than 256 bases go to the wavefront-per-read kernel (short-read rule up to 500 bases, long-read rule beyond)
        uint32_t max_len = 1, max_long = 0;
        std::vector<uint32_t> long_list;
        for (uint64_t i = 0; i < n; i++) {
            const uint32_t l1 = h->mate_len[2 * i], l2 = paired ? h->mate_len[2 * i + 1] : 0u;
            if (!paired && l1 > 256) { long_list.push_back(static_cast<uint32_t>(i)); max_long = std::max(max_long, l1); }
            else max_len = std::max(max_len, std::max(l1, l2));
        }
        if (max_len > 256) fail(LCTY_ERR_UNSUPPORTED, "recruitment: a mate of %u bases in a read pair (the device kernel takes mates of up to 256 bases)", max_len);
        if (n > 0xFFFFFFFFull) fail(LCTY_ERR_UNSUPPORTED, "recruitment: more than 2^32 reads in one call");
        V.scratch_cap = max_len;                                                 // a mate has fewer minimizers than bases
        const uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>((n + 63) / 64, static_cast<uint64_t>(ctx->props.multiProcessorCount) * 16));
        DevBuf<uint64_t> d_scratch, d_long_h; DevBuf<uint8_t> d_long_f, d_scratch_f; DevBuf<uint32_t> d_long_list;
        d_scratch.alloc(static_cast<size_t>(blocks) * V.scratch_cap * 64); d_scratch_f.alloc(static_cast<size_t>(blocks) * V.scratch_cap * 64);
        V.scratch = d_scratch.p; V.scratch_f = d_scratch_f.p;
        V.match_frac = t->prm.match_frac;
        // Params::new (recruit.rs:92-98)
        V.stretch_minims = (2 * t->prm.match_length + (static_cast<uint32_t>(t->prm.minimizer_w) + 1) - 1) / (static_cast<uint32_t>(t->prm.minimizer_w) + 1);
        V.stretch_score = static_cast<uint32_t>(std::ceil(std::max(static_cast<double>(V.stretch_minims) * (4.0 * t->prm.match_frac - 1.0), 3.0)));
        ctx->timed(LCTY_K_RECRUIT, [&] {
            hipLaunchKernelGGL(recruit_kernel, dim3(blocks), dim3(64), lds, s, V);
        });
        if (!long_list.empty()) {
            V.n_long = static_cast<uint32_t>(long_list.size()); V.long_cap = max_long + 64;
            const uint32_t lblocks = static_cast<uint32_t>(std::min<uint64_t>(long_list.size(), static_cast<uint64_t>(ctx->props.multiProcessorCount) * 8));
            d_long_list.alloc(long_list.size()); d_long_list.upload(long_list.data(), long_list.size(), s);
            d_long_h.alloc(static_cast<size_t>(lblocks) * 2 * V.long_cap); d_long_f.alloc(static_cast<size_t>(lblocks) * 2 * V.long_cap);
            V.long_list = d_long_list.p; V.long_h = d_long_h.p; V.long_f = d_long_f.p;
            ctx->timed(LCTY_K_RECRUIT, [&] {
                hipLaunchKernelGGL(recruit_single_kernel, dim3(lblocks), dim3(64), lds_single, s, V);
            });
        }
        LCTY_HIP(hipGetLastError());
        uint32_t err = 0;
        t->d_err.download(&err, 1, s);
        d_cnt.download(out_cnt, n, s); d_loci.download(out_loci, n * max_out, s);
        LCTY_HIP(hipStreamSynchronize(s));
        if (err == LCTY_ERR_UNSUPPORTED) fail(LCTY_ERR_UNSUPPORTED, "recruitment: a read matching minimizers of more than %u loci (%u for single reads beyond 256 bases)", MAX_LOCI_PER_READ, LONG_LOCI);
        if (err == LCTY_ERR_INVALID_INPUT) fail(LCTY_ERR_INVALID_INPUT, "recruitment: max_out is smaller than the number of loci of a read");
        if (err) fail(static_cast<int32_t>(err), "recruitment failed on the device");
        for (uint64_t i = 0; i < n; i++) std::sort(out_loci + i * max_out, out_loci + i * max_out + out_cnt[i]);
    });
}

}  // extern "C"
