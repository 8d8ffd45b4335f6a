// lcty_map_index.hip — the k-mer index of the basis alleles (lcty_locus_build_map_index), built on the device.
//
// What the mapper's kernels look up (lcty_map.hip, lcty_map_long.hip): an open-addressing table canonical k-mer -> run of places,
// a place = basis allele << 33 | position << 1 | (the forward k-mer is the canonical one), the run of a k-mer in (basis allele,
// position) order. Round 3 made it on the host (a sort of 12.8 M pairs for 256 alleles of 50 kb: 1.1 s per locus, 2.5 times the
// mapping call of 2 048 long reads). Here: one thread per window writes (k-mer, place) — windows with a base other than ACGT get the
// key ~0 and sort behind everything —, a stable radix sort by k-mer (below; the places were written in (allele, position) order and
// stay so inside a run), heads of runs by comparison with the left neighbour, their starts by a prefix sum, and one thread per run
// claims a slot (64-bit compare-and-swap on the key, linear probing). Which slot a k-mer ends up in depends on who gets there first;
// what a lookup finds does not.
//
// The sort: least significant digit first, eight bits a pass, over the bytes a k-mer of this k can differ in plus the top byte (where
// the key of a window without a k-mer differs from all of them). A pass is three steps over tiles of 4 096 pairs, a wavefront per tile:
// the tile's count of every digit value (LDS atomics) into a [256][tiles] matrix; the exclusive prefix sums of that matrix in
// (digit, tile) order — the first place in the output of every (digit value, tile) —; and the scatter, in which the wavefront goes
// over its tile 64 pairs at a time, in order: the lanes that hold the same digit value find each other by eight ballots, the first of
// them takes their places from the tile's running counter of that value (LDS), a lane's place is that plus its rank among its peers.
// Pairs with equal digits keep their order: stable. The prefix sums are the file's own three-kernel scan (chunks of 4 096, their sums
// scanned by the same code one level up, added back).

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

// ---- exclusive prefix sums of 32-bit counts: out[i] = in[0] + .. + in[i - 1]
constexpr uint32_t SCAN_CHUNK = 4096;          // entries per workgroup: 256 threads x 16
__global__ __launch_bounds__(256) void scan_chunk_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n, uint32_t* __restrict__ sums) {
    __shared__ uint32_t wsum[4];
    const uint64_t first = static_cast<uint64_t>(blockIdx.x) * SCAN_CHUNK + threadIdx.x * 16ull;
    uint32_t v[16], mine = 0;
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) { v[j] = first + j < n ? in[first + j] : 0u; mine += v[j]; }
    uint32_t incl = mine;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(incl, o); if ((threadIdx.x & 63u) >= static_cast<uint32_t>(o)) incl += up; }
    if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = incl - mine;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) before += wsum[w];
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) { if (first + j < n) out[first + j] = before; before += v[j]; }
    if (threadIdx.x == 255) sums[blockIdx.x] = before;
}
__global__ __launch_bounds__(256) void scan_add_kernel(uint32_t* __restrict__ out, uint64_t n, const uint32_t* __restrict__ chunk_before) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) out[i] += chunk_before[i / SCAN_CHUNK];
}
// scratch: at least scan_scratch_words(n) words
size_t scan_scratch_words(uint64_t n) {
    size_t words = 0;
    while (n > 1) { n = (n + SCAN_CHUNK - 1) / SCAN_CHUNK; words += 2 * n; if (n == 1) break; }
    return words + 2;
}
void exclusive_scan(const uint32_t* in, uint32_t* out, uint64_t n, uint32_t* scratch, hipStream_t s) {
    if (!n) return;
    const uint64_t chunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    uint32_t* sums = scratch, * before = scratch + chunks;
    hipLaunchKernelGGL(scan_chunk_kernel, dim3(static_cast<uint32_t>(chunks)), dim3(256), 0, s, in, out, n, sums);
    if (chunks > 1) {
        exclusive_scan(sums, before, chunks, scratch + 2 * chunks, s);
        hipLaunchKernelGGL(scan_add_kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, s, out, n, before);
    }
    LCTY_HIP(hipGetLastError());
}

// ---- one pass of the stable radix sort of (key, value) pairs by the eight bits of the key from `shift` up
constexpr uint32_t SORT_TILE = 4096;           // pairs per wavefront
__global__ __launch_bounds__(64) void sort_count_kernel(const uint64_t* __restrict__ keys, uint64_t n, uint32_t shift, uint32_t* __restrict__ counts,
                                                        uint32_t n_tiles) {
    __shared__ uint32_t h[256];
    const uint32_t lane = threadIdx.x;
    for (uint32_t d = lane; d < 256; d += 64) h[d] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const uint64_t base = static_cast<uint64_t>(blockIdx.x) * SORT_TILE;
    for (uint32_t r = 0; r < SORT_TILE / 64; r++) {
        const uint64_t i = base + r * 64ull + lane;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 0xFFu], 1u);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t d = lane; d < 256; d += 64) counts[static_cast<size_t>(d) * n_tiles + blockIdx.x] = h[d];
}
__global__ __launch_bounds__(64) void sort_scatter_kernel(const uint64_t* __restrict__ keys_in, const uint64_t* __restrict__ vals_in,
                                                          uint64_t* __restrict__ keys_out, uint64_t* __restrict__ vals_out, uint64_t n, uint32_t shift,
                                                          const uint32_t* __restrict__ first, uint32_t n_tiles) {
    __shared__ uint32_t next[256];             // where the tile's next pair of every digit value goes
    const uint32_t lane = threadIdx.x;
    for (uint32_t d = lane; d < 256; d += 64) next[d] = first[static_cast<size_t>(d) * n_tiles + blockIdx.x];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const uint64_t base = static_cast<uint64_t>(blockIdx.x) * SORT_TILE;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t r = 0; r < SORT_TILE / 64; r++) {
        const uint64_t i = base + r * 64ull + lane;
        const bool valid = i < n;
        const uint64_t k = valid ? keys_in[i] : 0ull, v = valid ? vals_in[i] : 0ull;
        const uint32_t d = static_cast<uint32_t>(k >> shift) & 0xFFu;
        unsigned long long peers = __ballot(valid);                      // the lanes of this step with my digit value
#pragma unroll
        for (uint32_t b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long vote = __ballot(valid && bit);
            peers &= bit ? vote : ~vote;
        }
        const uint32_t rank = static_cast<uint32_t>(__popcll(peers & below));
        const int leader = valid ? __ffsll(static_cast<long long>(peers)) - 1 : static_cast<int>(lane);
        uint32_t at = 0;
        if (valid && static_cast<int>(lane) == leader) { at = next[d]; next[d] = at + static_cast<uint32_t>(__popcll(peers)); }
        at = static_cast<uint32_t>(__shfl(static_cast<int>(at), leader));
        if (valid) { keys_out[at + rank] = k; vals_out[at + rank] = v; }
    }
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
    if (n_windows > 0x7FFFFFF0ull) fail(LCTY_ERR_UNSUPPORTED, "more than 2^31 k-mer places in the basis alleles");   // places and run starts are 32-bit; the sort's offsets too
    auto ix = std::make_shared<MapIndex>();
    ix->basis.alloc(n_basis); ix->basis.upload(basis, n_basis, s);
    ix->k = k; ix->n_basis = n_basis;
    DevBuf<uint64_t> d_win_off, keys_a, keys_b, places_a;
    DevBuf<unsigned long long> d_counts;
    DevBuf<uint32_t> head, rank, run_start, tile_counts, tile_first, scan_tmp;
    d_win_off.alloc(n_basis + 1); d_win_off.upload(win_off.data(), n_basis + 1, s);
    d_counts.alloc(1); d_counts.zero(s);
    const size_t n = static_cast<size_t>(std::max<uint64_t>(n_windows, 1));
    keys_a.alloc(n); keys_b.alloc(n); places_a.alloc(n);
    ix->entries.alloc(n);
    unsigned long long n_invalid = 0;
    uint32_t n_valid = 0, n_runs = 0;
    if (n_windows) {
        // the bytes of the key that matter: those below bit 2 k, and the top one (the key of a window without a k-mer is ~0)
        std::vector<uint32_t> shifts;
        for (uint32_t b = 0; b < 8 && 8 * b < 2 * k; b++) shifts.push_back(8 * b);
        if (shifts.back() != 56) shifts.push_back(56);
        // the pairs are written where an even number of passes away from (keys_b, entries): the last pass ends there
        uint64_t* ka = keys_a.p; uint64_t* va = places_a.p; uint64_t* kb = keys_b.p; uint64_t* vb = ix->entries.p;
        if (shifts.size() % 2 == 0) { std::swap(ka, kb); std::swap(va, vb); }
        const uint32_t blocks = static_cast<uint32_t>((n_windows + 255) / 256);
        hipLaunchKernelGGL(index_pairs_kernel, dim3(blocks), dim3(256), 0, s, locus->d_seqs.p, locus->d_seq_off.p, ix->basis.p, d_win_off.p, n_basis, k,
                           n_windows, ka, va, d_counts.p);
        LCTY_HIP(hipGetLastError());
        const uint32_t n_tiles = static_cast<uint32_t>((n_windows + SORT_TILE - 1) / SORT_TILE);
        const uint64_t cells = 256ull * n_tiles;
        tile_counts.alloc(cells); tile_first.alloc(cells); scan_tmp.alloc(scan_scratch_words(cells));
        for (uint32_t shift : shifts) {
            hipLaunchKernelGGL(sort_count_kernel, dim3(n_tiles), dim3(64), 0, s, ka, n_windows, shift, tile_counts.p, n_tiles);
            exclusive_scan(tile_counts.p, tile_first.p, cells, scan_tmp.p, s);
            hipLaunchKernelGGL(sort_scatter_kernel, dim3(n_tiles), dim3(64), 0, s, ka, va, kb, vb, n_windows, shift, tile_first.p, n_tiles);
            LCTY_HIP(hipGetLastError());
            std::swap(ka, kb); std::swap(va, vb);
        }
        d_counts.download(&n_invalid, 1, s);
        LCTY_HIP(hipStreamSynchronize(s));
        n_valid = static_cast<uint32_t>(n_windows - n_invalid);
    }
    if (n_valid) {
        const uint32_t blocks = (n_valid + 255) / 256;
        head.alloc(n_valid); rank.alloc(n_valid);
        hipLaunchKernelGGL(index_heads_kernel, dim3(blocks), dim3(256), 0, s, keys_b.p, static_cast<uint64_t>(n_valid), head.p);
        LCTY_HIP(hipGetLastError());
        if (scan_tmp.n < scan_scratch_words(n_valid)) scan_tmp.alloc(scan_scratch_words(n_valid));
        exclusive_scan(head.p, rank.p, n_valid, scan_tmp.p, s);
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
