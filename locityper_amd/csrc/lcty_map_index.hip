// lcty_map_index.hip — the k-mer index of the basis alleles (lcty_locus_build_map_index), built on the device.
//
// What the mapper's kernels look up (lcty_map.hip, lcty_map_long.hip): an open-addressing table canonical k-mer -> run of places,
// a place = basis allele << 33 | position << 1 | (the forward k-mer is the canonical one), the run of a k-mer in (basis allele,
// position) order. Round 3 made it on the host (a sort of 12.8 M pairs for 256 alleles of 50 kb: 1.1 s per locus, 2.5 times the
// mapping call of 2 048 long reads). Here: one thread per window writes (k-mer, place) — windows with a base other than ACGT get the
// key ~0 and sort behind everything —, a stable radix sort by k-mer (rocPRIM through hipCUB; the places were written in (allele,
// position) order and stay so inside a run), heads of runs by comparison with the left neighbour, their starts by a prefix sum, and
// one thread per run claims a slot (64-bit compare-and-swap on the key, linear probing). Which slot a k-mer ends up in depends on who
// gets there first; what a lookup finds does not.
#include <hipcub/hipcub.hpp>

#include "lcty_common.hpp"
#include "lcty_map_internal.hpp"

namespace lcty {
namespace {

// window w of the concatenated windows of the basis alleles: win_off[b] <= w < win_off[b + 1]
__global__ __launch_bounds__(256) void index_pairs_kernel(const uint8_t* __restrict__ seqs, const uint64_t* __restrict__ seq_off,
                                                          const uint16_t* __restrict__ basis, const uint64_t* __restrict__ win_off, uint32_t n_basis,
                                                          uint32_t k, uint64_t n_windows, uint64_t* __restrict__ keys, uint64_t* __restrict__ places,
                                                          unsigned long long* __restrict__ n_invalid) {
    const uint64_t w = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
    uint32_t bad = 0;
    if (w < n_windows) {
        uint32_t lo = 0, hi = n_basis;                               // the allele of this window
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (win_off[mid] <= w) lo = mid; else hi = mid; }
        const uint32_t b = lo;
        const uint64_t i = w - win_off[b];
        const uint8_t* s = seqs + seq_off[basis[b]] + i;
        uint64_t fw = 0, rv = 0;
        bool ok = true;
        for (uint32_t j = 0; j < k; j++) {
            const uint32_t e = enc_of(s[j]);
            ok &= e < 4u;
            fw = (fw << 2) | (e & 3u);
            rv = (rv >> 2) | (static_cast<uint64_t>(3u - (e & 3u)) << (2 * k - 2));
        }
        const bool fwd = fw <= rv;
        keys[w] = ok ? (fwd ? fw : rv) : MAP_FREE;
        places[w] = (static_cast<uint64_t>(b) << 33) | (i << 1) | (fwd ? 1ull : 0ull);
        bad = ok ? 0u : 1u;
    }
    const unsigned long long m = __ballot(bad != 0);
    if ((threadIdx.x & 63u) == 0 && m) atomicAdd(n_invalid, static_cast<unsigned long long>(__popcll(m)));
}

__global__ __launch_bounds__(256) void index_heads_kernel(const uint64_t* __restrict__ keys, uint64_t n, uint32_t* __restrict__ head) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) head[i] = i == 0 || keys[i] != keys[i - 1] ? 1u : 0u;
}

// run r starts at the element whose exclusive prefix sum of heads is r and that is a head
__global__ __launch_bounds__(256) void index_starts_kernel(const uint32_t* __restrict__ head, const uint32_t* __restrict__ rank, uint64_t n,
                                                           uint32_t* __restrict__ run_start) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i < n && head[i]) run_start[rank[i]] = static_cast<uint32_t>(i);
}

__global__ __launch_bounds__(256) void index_insert_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ run_start, uint32_t n_runs,
                                                           uint32_t n_valid, MapSlot* __restrict__ table, uint64_t mask) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_runs) return;
    const uint32_t start = run_start[r], end = r + 1 < n_runs ? run_start[r + 1] : n_valid;
    const uint64_t key = keys[start];
    uint64_t h = map_hash(key) & mask;
    for (;;) {
        const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&table[h].key), static_cast<unsigned long long>(MAP_FREE),
                                                 static_cast<unsigned long long>(key));
        if (old == MAP_FREE) break;                                  // every k-mer has one run: nobody else asks for this key
        h = (h + 1) & mask;
    }
    table[h].start = start; table[h].count = end - start;
}

}  // namespace

std::shared_ptr<MapIndex> build_map_index_device(lcty_locus* locus, const uint16_t* basis, uint32_t n_basis, uint32_t k) {
    lcty_ctx* ctx = locus->ctx;
    hipStream_t s = ctx->stream;
    std::vector<uint64_t> seq_off(locus->n_alleles + 1);
    locus->d_seq_off.download(seq_off.data(), seq_off.size(), s);
    LCTY_HIP(hipStreamSynchronize(s));
    std::vector<uint64_t> win_off(n_basis + 1, 0);
    for (uint32_t b = 0; b < n_basis; b++) {
        const uint64_t len = seq_off[basis[b] + 1] - seq_off[basis[b]];
        win_off[b + 1] = win_off[b] + (len >= k ? len + 1 - k : 0);
    }
    const uint64_t n_windows = win_off[n_basis];
    if (n_windows > 0x7FFFFFF0ull) fail(LCTY_ERR_UNSUPPORTED, "more than 2^31 k-mer places in the basis alleles");
    auto ix = std::make_shared<MapIndex>();
    ix->basis.alloc(n_basis); ix->basis.upload(basis, n_basis, s);
    ix->k = k; ix->n_basis = n_basis;
    DevBuf<uint64_t> d_win_off, keys_a, keys_b, places_a;
    DevBuf<unsigned long long> d_counts;
    DevBuf<uint32_t> head, rank, run_start;
    DevBuf<uint8_t> tmp;
    d_win_off.alloc(n_basis + 1); d_win_off.upload(win_off.data(), n_basis + 1, s);
    d_counts.alloc(1); d_counts.zero(s);
    const size_t n = static_cast<size_t>(std::max<uint64_t>(n_windows, 1));
    keys_a.alloc(n); keys_b.alloc(n); places_a.alloc(n);
    ix->entries.alloc(n);
    unsigned long long n_invalid = 0;
    uint32_t n_valid = 0, n_runs = 0;
    if (n_windows) {
        const uint32_t blocks = static_cast<uint32_t>((n_windows + 255) / 256);
        hipLaunchKernelGGL(index_pairs_kernel, dim3(blocks), dim3(256), 0, s, locus->d_seqs.p, locus->d_seq_off.p, ix->basis.p, d_win_off.p, n_basis, k,
                           n_windows, keys_a.p, places_a.p, d_counts.p);
        LCTY_HIP(hipGetLastError());
        size_t tb = 0;
        LCTY_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, keys_a.p, keys_b.p, places_a.p, ix->entries.p, static_cast<int>(n_windows), 0, 64, s));
        tmp.alloc(std::max<size_t>(tb, 1));
        LCTY_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, keys_a.p, keys_b.p, places_a.p, ix->entries.p, static_cast<int>(n_windows), 0, 64, s));
        d_counts.download(&n_invalid, 1, s);
        LCTY_HIP(hipStreamSynchronize(s));
        n_valid = static_cast<uint32_t>(n_windows - n_invalid);
    }
    if (n_valid) {
        const uint32_t blocks = (n_valid + 255) / 256;
        head.alloc(n_valid); rank.alloc(n_valid);
        hipLaunchKernelGGL(index_heads_kernel, dim3(blocks), dim3(256), 0, s, keys_b.p, static_cast<uint64_t>(n_valid), head.p);
        size_t tb = 0;
        LCTY_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, head.p, rank.p, static_cast<int>(n_valid), s));
        if (tmp.n < tb) tmp.alloc(tb);
        LCTY_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, head.p, rank.p, static_cast<int>(n_valid), s));
        uint32_t last_rank = 0, last_head = 0;
        rank.download(&last_rank, 1, s, n_valid - 1); head.download(&last_head, 1, s, n_valid - 1);
        LCTY_HIP(hipStreamSynchronize(s));
        n_runs = last_rank + last_head;
        run_start.alloc(n_runs);
        hipLaunchKernelGGL(index_starts_kernel, dim3(blocks), dim3(256), 0, s, head.p, rank.p, static_cast<uint64_t>(n_valid), run_start.p);
        LCTY_HIP(hipGetLastError());
    }
    uint64_t cap = 1024;
    while (cap < 2 * static_cast<uint64_t>(n_runs)) cap <<= 1;
    ix->table.alloc(cap);
    LCTY_HIP(hipMemsetAsync(ix->table.p, 0xFF, cap * sizeof(MapSlot), s));          // key ~0 = free (start / count of a free slot are never read)
    ix->mask = cap - 1;
    if (n_runs) {
        hipLaunchKernelGGL(index_insert_kernel, dim3((n_runs + 255) / 256), dim3(256), 0, s, keys_b.p, run_start.p, n_runs, n_valid, ix->table.p, ix->mask);
        LCTY_HIP(hipGetLastError());
    }
    LCTY_HIP(hipStreamSynchronize(s));
    return ix;
}

}  // namespace lcty
