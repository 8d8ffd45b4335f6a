// lcty_device.hpp — plain views passed by value to the gfx950 kernels, and the
// device helpers shared between kernels (2-bit k-mer extraction, hash probing).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/locityper_hip.h"

namespace lcty {

constexpr int WAVE = 64;                       // CDNA4 wavefront
// the pair-alignment arena of a scored batch: a wavefront of a large scoring launch reserves PA_CHUNK entries at a time
// (lcty_score.hip); a batch that can hold such a launch gets an eighth more room and a chunk per wavefront (lcty_reads_create)
constexpr uint32_t PA_CHUNK = 8192, PA_POOL_MIN_PAIRS = 32, PA_MAX_GRID = 256 * 16;
constexpr uint64_t KSET_EMPTY = ~0ull;         // never a valid canonical k-mer for k <= 31

// Device record of one PairAlignment (src/model/locs.rs:668-676), 24 B.
struct PairAlnDev {
    double ln_prob;
    uint32_t mid1, mid2;     // LCTY_NONE_U32 = None
    uint16_t contig;
    uint16_t ix1, ix2;       // record index inside the pair, 0xFFFF = None
    uint16_t _pad;
};
static_assert(sizeof(PairAlnDev) == 24, "PairAlnDev layout");

// BayesCalc<NBinom, NBinom> of one GC bin (src/model/distr_cache.rs:61-75): null hypothesis + alternatives, for the
// direct evaluation of depths beyond the 256-entry LinearCache (src/math/distr/lincache.rs:41-48).
struct DepthNB {
    double lnq;                                  // ln(1 - p), shared by all hypotheses (NBinom::mul keeps p)
    double n[LCTY_MAX_ALT_CN + 1];               // [0] = null (cn 1), then the alternatives
    double lnpmf_const[LCTY_MAX_ALT_CN + 1];     // n ln p - lnGamma(n)
};

struct LocusView {
    uint32_t n_alleles, k;
    const uint32_t* allele_len;     // [A]
    const uint32_t* ci_off;         // [A+1] offsets into the per-position ContigInfo arrays
    const uint16_t* compl_cnt;      // linguistic-complexity numerators (src/seq/compl.rs:115-140)
    double compl_mult;              // 1 / min(neighb+1-ck, 4^ck)
    uint32_t half_neighb;
    // UniqueKmers (src/model/locs.rs:915-1003)
    const uint64_t* kset;           // open addressing, KSET_EMPTY = free
    uint64_t kset_mask;             // capacity - 1
    uint32_t undef_in_set;          // UNDEF was inserted (an allele window with N has count 0)
    double weight_mult, weight_interc;
    // InsertDistr (src/bg/insertsz.rs)
    const double* ins_lut;
    uint32_t ins_lut_size;
    double ins_n, ins_lnq, ins_lnpmf_const, insert_penalty;
    // ErrorProfile (src/bg/err_prof.rs:212-221)
    double lp[5];
    // EditDistCache (src/bg/err_prof.rs:415-448): (good, passable) per read length
    const uint2* edit_lut;
    uint32_t edit_lut_size;
    // model::Params
    double unmapped_penalty, prob_diff, min_weight, poor_compl, poor_compl_edit;
    uint32_t boundary;              // boundary_size - tweak (locs.rs:1099)
    uint32_t is_paired, short_reads, strict_subset;
    // ExplicitWeights (src/model/windows.rs:196-250), null without --reg-weights: len + 1 values per allele
    const double* ew_val;
    const uint64_t* ew_off;         // [A+1]
    uint32_t half_window;           // window_size / 2 (windows.rs:499)
};

struct ReadsView {
    uint64_t n_pairs;
    const uint32_t* mate_len;
    const uint64_t* mate_off;
    const uint32_t* bases2;
    const uint32_t* nmask;
    const uint64_t* aln_off;
    const lcty_aln_rec* recs;
    const uint64_t* cigar_off;
    const uint32_t* cigar;
    const uint2* pair_meta;         // per pair {index of the mate-2 primary (or n), number of records to look at}
    // products of AllAlignments::load
    uint8_t* status;
    double* weight;
    double* unmapped_prob;
    uint16_t* uniq_kmers;           // [2R]
    double* matrix;                 // [R][A] read-major; rows of non-GOOD pairs are 0.0
    PairAlnDev* pa;                 // arena
    uint64_t pa_cap;
    uint32_t pa_chunk;              // entries a wavefront of the scoring kernel reserves at a time (0: every pair reserves its own)
    unsigned long long* pa_count;   // arena cursor
    uint64_t* pa_off;               // [R]
    uint32_t* pa_cnt;               // [R]
    uint32_t* pa_idx;               // [R][A]: offset of the contig's entries inside the pair's arena segment | count << 24
    uint32_t* err_flag;             // first LCTY_ERR_* raised by a kernel
    double* recover_w;              // per pair: read weight if it reaches recover_and_group_alignments (locs.rs:1255), else -1
    // the lean scoring kernel hands the pairs it does not take (several saved alignments of a read end on one contig, mates
    // beyond the register path of the k-mer windows) to the general kernel: their indices, appended in any order
    uint32_t* defer_list;           // [cap_raw_pairs] or null
    unsigned int* defer_count;      // [1]
    const uint32_t* only_list;      // general kernel: the pairs to score (null: all n_pairs) ...
    const unsigned int* only_count; // ... and how many (device memory: no host round trip between the two launches)
    unsigned long long* dbg;        // lcty_ctx_set_knob "score_timing": shader-clock sums of the lean kernel's phases (else null)
    uint8_t* park;                  // scoring kernel, large pairs: per-workgroup scratch of saved alignments (else null)
    uint64_t park_stride;
};

__host__ __device__ inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

#ifdef __HIPCC__
// Canonical k-mer (src/seq/kmers.rs:192-196) of the window starting at base q of a mate whose
// 2-bit stream starts at 64-bit word `w64` (mate offsets are multiples of 32 bases).
// The stream is LSB-first: x = sum enc[q+t] << 2t, hence rv = ~x (masked) and fw = digit-reverse(x).
__device__ inline uint64_t canonical_kmer_2bit(const uint64_t* w64, uint32_t q, uint32_t k) {
    const uint32_t word = q >> 5, sh = (q & 31u) * 2u;
    uint64_t x = w64[word] >> sh;
    if (sh + 2u * k > 64u) x |= w64[word + 1] << (64u - sh);
    const uint64_t mask = (1ull << (2u * k)) - 1ull;
    x &= mask;
    const uint64_t rv = (~x) & mask;
    uint64_t y = __brevll(x);
    y = ((y >> 1) & 0x5555555555555555ull) | ((y & 0x5555555555555555ull) << 1);
    const uint64_t fw = y >> (64u - 2u * k);
    return rv < fw ? rv : fw;
}

// any "not ACGT" base inside [q, q+k) of the mate's 1-bit stream starting at 32-bit word `nm`
__device__ inline bool window_has_n(const uint32_t* nm, uint32_t q, uint32_t k) {
    const uint32_t w = q >> 5, s = q & 31u;
    uint32_t bits = nm[w] >> s;
    if (s + k > 32u) bits |= nm[w + 1] << (32u - s);
    return (bits & ((1u << k) - 1u)) != 0u;
}

// ---- k-mers of 32..63 bases (the reference keeps every k-mer of UniqueKmers in a u128, locs.rs:919-963): 128-bit keys as {lo, hi}
// pairs, the set an open-addressing table of pairs built on the host (lcty_locus_create), free = {~0, ~0} ----
struct Kmer128 { uint64_t lo, hi; };
__host__ __device__ inline uint64_t kmer128_hash(uint64_t lo, uint64_t hi) { return mix64(lo ^ mix64(hi ^ 0x9E3779B97F4A7C15ull)); }
__device__ __forceinline__ uint64_t pair_reverse64(uint64_t x) {
    const uint64_t y = __brevll(x);
    return ((y >> 1) & 0x5555555555555555ull) | ((y & 0x5555555555555555ull) << 1);
}
// canonical k-mer of the window at base q, 32 <= k <= 63 (the 64-bit form above, on two words)
__device__ inline Kmer128 canonical_kmer_2bit128(const uint64_t* w64, uint32_t q, uint32_t k) {
    const uint32_t word = q >> 5, sh = (q & 31u) * 2u, last = (q + k - 1) >> 5;
    const uint64_t a = w64[word], b = last > word ? w64[word + 1] : 0ull, c = last > word + 1 ? w64[word + 2] : 0ull;
    uint64_t xlo = a, xhi = b;
    if (sh) { xlo = (a >> sh) | (b << (64u - sh)); xhi = (b >> sh) | (c << (64u - sh)); }
    const uint32_t hb = 2u * k - 64u;                                    // bits of the k-mer in the high word: 0..62
    const uint64_t hmask = (1ull << hb) - 1ull;
    xhi &= hmask;
    const uint64_t rlo = ~xlo, rhi = (~xhi) & hmask;                      // the reverse complement's value
    const uint64_t ylo = pair_reverse64(xhi), yhi = pair_reverse64(xlo); // the digits of x in reverse order, at the top of 128 bits
    const uint32_t s = 128u - 2u * k;                                     // 2..64
    const uint64_t flo = s == 64u ? yhi : (ylo >> s) | (yhi << (64u - s)), fhi = s == 64u ? 0ull : yhi >> s;
    const bool rv_less = rhi < fhi || (rhi == fhi && rlo < flo);
    return rv_less ? Kmer128{rlo, rhi} : Kmer128{flo, fhi};
}
// any "not ACGT" base inside [q, q+k), k <= 63
__device__ inline bool window_has_n_wide(const uint32_t* nm, uint32_t q, uint32_t k) {
    const uint32_t w = q >> 5, s = q & 31u, last = (q + k - 1) >> 5;
    uint64_t bits = (static_cast<uint64_t>(nm[w]) | (last > w ? static_cast<uint64_t>(nm[w + 1]) << 32 : 0ull)) >> s;
    if (last > w + 1) bits |= static_cast<uint64_t>(nm[w + 2]) << (64u - s);       // s > 0 here: three words only with an offset
    return (bits & ((1ull << k) - 1ull)) != 0ull;
}
__device__ inline bool kset128_contains(const uint64_t* kset, uint64_t mask, Kmer128 key) {
    uint64_t slot = kmer128_hash(key.lo, key.hi) & mask;
    while (true) {
        const uint64_t lo = kset[2 * slot], hi = kset[2 * slot + 1];
        if (lo == key.lo && hi == key.hi) return true;
        if (lo == KSET_EMPTY && hi == KSET_EMPTY) return false;
        slot = (slot + 1) & mask;
    }
}

__device__ inline bool kset_contains(const uint64_t* kset, uint64_t mask, uint64_t key) {
    uint64_t slot = mix64(key) & mask;
    while (true) {
        const uint64_t v = kset[slot];
        if (v == key) return true;
        if (v == KSET_EMPTY) return false;
        slot = (slot + 1) & mask;
    }
}
#endif

}  // namespace lcty
