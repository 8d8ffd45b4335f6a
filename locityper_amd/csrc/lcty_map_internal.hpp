// lcty_map_internal.hpp — what the two routes of candidate generation share (lcty_map.hip: read ends of up to 256 bases on up to 32
// basis alleles; lcty_map_long.hip: read ends of any length on up to 256 basis alleles): the k-mer index of the basis alleles, the
// packed-base accessors and the device buffers of one call.
#pragma once

#include <memory>

#include "lcty_objects.hpp"

namespace lcty {

constexpr uint32_t MAP_MAX_LEN = 256;          // bases per read end on the short route
constexpr uint32_t MAP_MAX_BASIS = 32;         // basis alleles on the short route
constexpr uint32_t MAP_MAX_BAND = 16;          // diagonals on either side in an alignment with gaps
constexpr uint32_t MAP_LONG_MAX_BASIS = 256;   // basis alleles on the long route
constexpr uint32_t MAP_LONG_MAX_LEN = (1u << 20) - 1;

struct MapSlot { uint64_t key; uint32_t start, count; };   // key ~0 = free
constexpr uint64_t MAP_FREE = ~0ull;

__host__ __device__ inline uint64_t map_hash(uint64_t x) {          // the mix of kmers.rs:93-103
    x = ~x; x ^= x >> 23; x *= 0x2127599bf4325c37ull; x ^= x >> 47;
    return x;
}

__device__ __forceinline__ uint32_t base_at(const uint32_t* b2, uint64_t off, uint32_t i) {
    const uint64_t p = off + i;
    return (b2[p >> 4] >> (2 * (p & 15u))) & 3u;
}
__device__ __forceinline__ bool n_at(const uint32_t* nm, uint64_t off, uint32_t i) {
    const uint64_t p = off + i;
    return (nm[p >> 5] >> (p & 31u)) & 1u;
}
__device__ __forceinline__ uint32_t enc_of(uint8_t c) { return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u; }

struct MapIndex {
    DevBuf<MapSlot> table; DevBuf<uint64_t> entries; DevBuf<uint16_t> basis; DevBuf<uint32_t> scratch;
    uint64_t mask = 0; uint32_t k = 0, n_basis = 0;
};

// lcty_map_index.hip: the index of the given basis alleles, made on the device
std::shared_ptr<MapIndex> build_map_index_device(lcty_locus* locus, const uint16_t* basis, uint32_t n_basis, uint32_t k);

// both passes of a route; the records stay on the device, the offsets come to the host
struct MapRun {
    DevBuf<uint32_t> d_len, d_b2, d_nm, d_nrec, d_ncig, d_ob2, d_onm, d_cigar, d_nhave, d_work, d_counters, d_ops;
    DevBuf<uint8_t> d_cands;                              // MapCand / LongCand records
    uint32_t max_rec_cigar = 0;
    DevBuf<uint64_t> d_off, d_rec_at, d_cig_at, d_pair_cig;
    DevBuf<lcty_aln_rec> d_recs;
    std::vector<uint32_t> nrec, ncig;
    uint64_t n_recs = 0, n_cigar = 0;
    // scratch of the long route
    DevBuf<uint4> d_anchors; DevBuf<uint2> d_chain; DevBuf<uint8_t> d_dirs; DevBuf<uint32_t> d_opsbuf;
};

// lcty_map_long.hip: the long route over an uploaded chunk (X.d_len .. X.d_nm filled); leaves records, CIGAR words, re-oriented bases
// and the per-read-end counts in X, the offsets in aln_off / cigar_off
void run_map_long(lcty_locus* locus, const lcty_reads_host* chunk, const lcty_map_params* params, const MapIndex& ix, uint32_t max_len,
                  uint64_t* aln_off, uint64_t* cigar_off, bool sizes_only, MapRun& X);

}  // namespace lcty
