// lcty_reads.hip — device-resident batch of read pairs: creation, chunked append (H2D),
// and the getters that hand the products of AllAlignments::load back to the host.
#include <algorithm>
#include <memory>

#include <string>
#include <system_error>
#include <thread>

#include "lcty_objects.hpp"

using namespace lcty;

namespace lcty {

// out[a * n_good + j] = M[good_ix[j]][a]  (AllAlignments::best_aln_matrix layout, locs.rs:1203-1212)
__global__ void compact_matrix_kernel(const double* __restrict__ M, const uint64_t* __restrict__ good_ix,
                                      uint64_t n_good, uint32_t A, double* __restrict__ out) {
    __shared__ double tile[64][65];
    const uint64_t j0 = static_cast<uint64_t>(blockIdx.x) * 64;
    const uint32_t a0 = blockIdx.y * 64;
    for (uint32_t t = threadIdx.y; t < 64; t += blockDim.y) {
        const uint64_t j = j0 + t;
        const uint32_t a = a0 + threadIdx.x;
        tile[t][threadIdx.x] = (j < n_good && a < A) ? M[good_ix[j] * A + a] : 0.0;
    }
    __syncthreads();
    for (uint32_t t = threadIdx.y; t < 64; t += blockDim.y) {
        const uint32_t a = a0 + t;
        const uint64_t j = j0 + threadIdx.x;
        if (a < A && j < n_good) out[static_cast<uint64_t>(a) * n_good + j] = tile[threadIdx.x][t];
    }
}

void compact_matrix(lcty_reads* reads, double* d_out, uint64_t n_good) {
    lcty_ctx* ctx = reads->ctx;
    std::vector<uint8_t> status(reads->n_pairs);
    reads->d_status.download(status.data(), reads->n_pairs, ctx->stream);
    LCTY_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<uint64_t> good;
    good.reserve(n_good);
    for (uint64_t r = 0; r < reads->n_pairs; r++) if (status[r] == LCTY_READ_GOOD) good.push_back(r);
    if (good.size() != n_good) fail(LCTY_ERR_RUNTIME, "inconsistent number of good reads");
    if (n_good == 0) return;
    DevBuf<uint64_t> d_good;
    d_good.alloc(n_good);
    d_good.upload(good.data(), n_good, ctx->stream);
    const uint32_t A = reads->locus->n_alleles;
    dim3 grid(static_cast<uint32_t>((n_good + 63) / 64), (A + 63) / 64);
    hipLaunchKernelGGL(compact_matrix_kernel, grid, dim3(64, 4), 0, ctx->stream, reads->d_matrix.p, d_good.p, n_good, A, d_out);
    LCTY_HIP(hipGetLastError());
    LCTY_HIP(hipStreamSynchronize(ctx->stream));
}


// ---- the indices of the GOOD pairs of a batch, in batch order (lcty_reads::ensure_good_index) ----
constexpr uint32_t GOOD_BLOCK = 1024;
__global__ __launch_bounds__(256) void good_count_kernel(const uint8_t* __restrict__ status, uint32_t n, uint32_t* __restrict__ cnt) {
    __shared__ uint32_t part[4];
    const uint32_t base = blockIdx.x * GOOD_BLOCK;
    uint32_t c = 0;
    for (uint32_t i = threadIdx.x; i < GOOD_BLOCK; i += 256) c += base + i < n && status[base + i] == LCTY_READ_GOOD;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) cnt[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
// cnt[0..blocks) -> exclusive prefix sums, cnt[blocks] = total (one workgroup of 1 024 threads, any number of blocks)
__global__ __launch_bounds__(1024) void good_scan_kernel(uint32_t* __restrict__ cnt, uint32_t blocks) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < blocks; b0 += 1024) {
        const uint32_t i = b0 + threadIdx.x;
        const uint32_t v = i < blocks ? cnt[i] : 0u;
        uint32_t incl = v;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(incl, o); if ((threadIdx.x & 63u) >= static_cast<uint32_t>(o)) incl += up; }
        if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t before = carry;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) before += wsum[w];
        if (i < blocks) cnt[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) cnt[blocks] = carry;
}
__global__ __launch_bounds__(256) void good_scatter_kernel(const uint8_t* __restrict__ status, uint32_t n, const uint32_t* __restrict__ cnt,
                                                           uint32_t* __restrict__ good_ix) {
    __shared__ uint32_t run;
    const uint32_t base = blockIdx.x * GOOD_BLOCK, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) run = cnt[blockIdx.x];
    __syncthreads();
    // the block's 1 024 pairs in order: wavefront w takes pairs [256 w, 256 w + 256) in four steps of 64, the wavefronts one after the other
    for (uint32_t w = 0; w < 4; w++) {
        if (wave == w) {
            uint32_t at = run;
            for (uint32_t step = 0; step < 4; step++) {
                const uint32_t i = base + w * 256 + step * 64 + lane;
                const bool good = i < n && status[i] == LCTY_READ_GOOD;
                const unsigned long long m = __ballot(good);
                if (good) good_ix[at + static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)))] = i;
                at += static_cast<uint32_t>(__popcll(m));
            }
            if (lane == 0) run = at;
        }
        __syncthreads();
    }
}
}  // namespace lcty

// The pairs whose records are on the device: all of them, or the current chunk of a streaming batch — then the per-pair
// products are addressed from raw_first on, the arena and its cursor are the batch's (pa_off stays absolute).
ReadsView lcty_reads::view() const {
    ReadsView v{};
    const uint64_t f = raw_first, A = locus->n_alleles;
    v.n_pairs = n_pairs - f;
    v.mate_len = d_mate_len.p; v.mate_off = d_mate_off.p; v.bases2 = d_bases2.p; v.nmask = d_nmask.p;
    v.aln_off = d_aln_off.p; v.recs = d_recs.p; v.cigar_off = d_cigar_off.p; v.cigar = d_cigar.p;
    v.pair_meta = d_pair_meta.p;
    v.status = d_status.p + f; v.weight = d_weight.p + f; v.unmapped_prob = d_unmapped.p + f; v.uniq_kmers = d_uniq.p + 2 * f;
    v.matrix = d_matrix.p + f * A;
    v.pa = d_pa.p; v.pa_cap = d_pa.n; v.pa_count = d_pa_count.p; v.pa_off = d_pa_off.p + f; v.pa_cnt = d_pa_cnt.p + f;
    v.pa_idx = d_pa_idx.p + f * A;
    v.err_flag = d_err.p;
    v.recover_w = d_recover_w.p + f;
    return v;
}

void lcty_reads::ensure_good_index() {
    if (good_valid) return;
    if (n_pairs >= 0xFFFFFFFFull) lcty::fail(LCTY_ERR_UNSUPPORTED, "more than 2^32 read pairs in one batch");
    // The indices of the GOOD pairs in batch order, made on the device (blocks of 1 024 pairs: count, scan of the counts by one workgroup,
    // ordered scatter): only their number comes back. The host loop over the status bytes (a megabyte down, four up, the loop itself) was
    // 3 ms on the critical path of every locus of a queue.
    hipStream_t s = ctx->stream;
    const uint32_t n = static_cast<uint32_t>(n_pairs), blocks = (n + lcty::GOOD_BLOCK - 1) / lcty::GOOD_BLOCK;
    d_good_ix.ensure(std::max<size_t>(n_pairs, 1));           // grow-only: a hipFree waits for every stream of the device
    d_good_cnt.ensure(static_cast<size_t>(blocks) + 1);
    uint32_t total = 0;
    if (n) {
        hipLaunchKernelGGL(lcty::good_count_kernel, dim3(blocks), dim3(256), 0, s, d_status.p, n, d_good_cnt.p);
        hipLaunchKernelGGL(lcty::good_scan_kernel, dim3(1), dim3(1024), 0, s, d_good_cnt.p, blocks);
        hipLaunchKernelGGL(lcty::good_scatter_kernel, dim3(blocks), dim3(256), 0, s, d_status.p, n, d_good_cnt.p, d_good_ix.p);
        LCTY_HIP(hipGetLastError());
        d_good_cnt.download(&total, 1, s, blocks);
        LCTY_HIP(hipStreamSynchronize(s));
    }
    n_good_cached = total;
    good_valid = true;
}

void lcty_reads::check_device_error(hipStream_t on) {
    uint32_t flag = 0;
    if (!on) on = ctx->stream;
    d_err.download(&flag, 1, on);
    LCTY_HIP(hipStreamSynchronize(on));
    if (flag == LCTY_ERR_INVALID_DATA)
        fail(LCTY_ERR_INVALID_DATA,
             "alignment table violates the input contract (unsupported CIGAR operation, hard-clipped or empty primary, "
             "missing mate, contig id out of range) — the reference panics here (aln.rs:311, locs.rs:509-526)");
    if (flag) fail(static_cast<int32_t>(flag), "pair-alignment arena overflow");
}

extern "C" {

// cap_pairs: pairs of the whole batch (products); raw_pairs / cap_bases / cap_recs / cap_cigar: what is resident at a time
static int32_t create_reads(lcty_locus* locus, uint64_t cap_pairs, uint64_t raw_pairs, uint64_t cap_bases, uint64_t cap_recs,
                            uint64_t cap_cigar, uint64_t cap_pair_alns, bool streaming, lcty_reads** out) {
    return guarded([&] {
        if (!locus || !out) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (cap_bases % 32) fail(LCTY_ERR_INVALID_INPUT, "cap_bases must be a multiple of 32");
        lcty_ctx* ctx = locus->ctx;
        ctx->activate();
        auto R = std::unique_ptr<lcty_reads>(new lcty_reads());
        R->locus = locus; R->ctx = ctx;
        R->cap_pairs = cap_pairs; R->cap_bases = cap_bases; R->cap_recs = cap_recs; R->cap_cigar = cap_cigar;
        R->streaming = streaming; R->cap_raw_pairs = raw_pairs;
        R->chunk_cap_recs = cap_recs; R->chunk_cap_cigar = cap_cigar;
        const uint32_t A = locus->n_alleles;
        R->d_mate_len.alloc(2 * raw_pairs);
        R->d_mate_off.alloc(2 * raw_pairs + 1);
        R->d_bases2.alloc(cap_bases / 16 + 4);      // +4 words: the k-mer window loader may touch one 64-bit word past a mate
        R->d_nmask.alloc(cap_bases / 32 + 2);
        R->d_aln_off.alloc(raw_pairs + 1);
        R->d_recs.alloc(std::max<uint64_t>(cap_recs, 1));
        R->d_cigar_off.alloc(raw_pairs + 1);
        R->d_cigar.alloc(cap_cigar + 16);            // +16: the CIGAR loader reads 8 words per record unconditionally
        R->d_pair_meta.alloc(std::max<uint64_t>(raw_pairs, 1));
        R->d_status.alloc(std::max<uint64_t>(cap_pairs, 1));
        R->d_weight.alloc(std::max<uint64_t>(cap_pairs, 1));
        R->d_unmapped.alloc(std::max<uint64_t>(cap_pairs, 1));
        R->d_uniq.alloc(std::max<uint64_t>(2 * cap_pairs, 1));
        R->d_matrix.alloc(std::max<uint64_t>(cap_pairs * A, 1));
        // every saved record yields at most one (aln, unmapped) entry plus its share of pairs; the kept
        // list is capped at MAX_USED_ALNS per (pair, contig with records) — bound by 10 per record is loose,
        // the tight bound is min(10 * contigs_with_records, pairs + alones) <= 2 * recs per typical data.
        // We allocate 10 per (pair, allele) capped by 2 * records + pairs and report overflow loudly.
        // A streaming batch does not know the records of the chunks to come: the caller says how many PairAlignments to
        // make room for, by default three per (pair, allele) (one per contig is the rule, 24 B each).
        const uint64_t pa_cap = streaming
            ? (cap_pair_alns ? cap_pair_alns : std::min<uint64_t>(static_cast<uint64_t>(LCTY_MAX_USED_ALNS), 3) * cap_pairs * A) + 64
            : std::min<uint64_t>(static_cast<uint64_t>(LCTY_MAX_USED_ALNS) * cap_pairs * A, 2 * cap_recs + cap_pairs) + 64;
        // large launches hand the arena out in chunks per wavefront (lcty_device.hpp: PA_CHUNK): room for what stays unused
        // (launches of at least PA_POOL_MIN_PAIRS pairs per wavefront: only batches that can hold that many pairs at a time; a
        // streaming batch whose caller said how many PairAlignments to make room for keeps exactly that room)
        uint64_t pa_cap_used = pa_cap;
        // lcty_ctx_set_knob "arena_cap_pct": that share of the bound (a caller that knows its data: one PairAlignment per (pair, allele)
        // is the rule, the bound is two per record; an arena that turns out too small fails loudly)
        if (!streaming && ctx->knob("arena_cap_pct", 0) > 0) pa_cap_used = std::max<uint64_t>(64, pa_cap / 100 * static_cast<uint64_t>(ctx->knob("arena_cap_pct", 0)));
        R->pa_pooled = raw_pairs >= static_cast<uint64_t>(PA_POOL_MIN_PAIRS) * PA_MAX_GRID / 2 && !(streaming && cap_pair_alns);
        const uint64_t pa_slack = R->pa_pooled ? pa_cap_used / 8 + static_cast<uint64_t>(PA_CHUNK) * PA_MAX_GRID : 0;
        R->d_pa.alloc(pa_cap_used + pa_slack);
        R->d_pa_count.alloc(1);
        R->d_pa_off.alloc(std::max<uint64_t>(cap_pairs, 1));
        R->d_pa_cnt.alloc(std::max<uint64_t>(cap_pairs, 1));
        R->d_pa_idx.alloc(std::max<uint64_t>(cap_pairs * A, 1));
        R->d_err.alloc(1);
        R->d_err.zero(ctx->stream);
        R->d_recover_w.alloc(std::max<uint64_t>(cap_pairs, 1));
        R->d_pa_count.zero(ctx->stream);
        const uint64_t zero = 0;
        R->d_mate_off.upload(&zero, 1, ctx->stream);
        R->d_aln_off.upload(&zero, 1, ctx->stream);
        R->d_cigar_off.upload(&zero, 1, ctx->stream);
        LCTY_HIP(hipMemsetAsync(R->d_bases2.p, 0, R->d_bases2.n * sizeof(uint32_t), ctx->stream));
        LCTY_HIP(hipMemsetAsync(R->d_nmask.p, 0, R->d_nmask.n * sizeof(uint32_t), ctx->stream));
        LCTY_HIP(hipStreamSynchronize(ctx->stream));
        *out = R.release();
    });
}

int32_t lcty_reads_create(lcty_locus* locus, uint64_t cap_pairs, uint64_t cap_bases, uint64_t cap_recs,
                          uint64_t cap_cigar, lcty_reads** out) {
    return create_reads(locus, cap_pairs, cap_pairs, cap_bases, cap_recs, cap_cigar, 0, false, out);
}

int32_t lcty_reads_create_streaming(lcty_locus* locus, uint64_t cap_pairs, uint64_t chunk_pairs, uint64_t chunk_bases,
                                    uint64_t chunk_recs, uint64_t chunk_cigar, uint64_t cap_pair_alns, lcty_reads** out) {
    if (chunk_pairs == 0 || chunk_pairs > cap_pairs) chunk_pairs = cap_pairs;
    return create_reads(locus, cap_pairs, chunk_pairs, chunk_bases, chunk_recs, chunk_cigar, cap_pair_alns, true, out);
}

// lcty_reads_append and lcty_reads_append_counted: `h` carries the raw records, or — counted != NULL — everything but records and
// CIGAR words, which `counted` (16-byte entries, one per record) replaces
// dev: the chunk's records, CIGAR words and (re-oriented) bases are on the device already (lcty_map.hip made them); h then gives the
// sequence layout and the record / CIGAR offsets only
static int32_t append_impl(lcty_reads* R, const lcty_reads_host* h, const lcty_aln_counted* counted, const lcty::DeviceRecords* dev = nullptr) {
    return guarded([&] {
        if (!R || !h) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (R->n_pairs > 0 && R->counted != (counted != nullptr))
            fail(LCTY_ERR_INVALID_INPUT, "a batch holds either BAM records or counted alignments, not both");
        lcty_ctx* ctx = R->ctx;
        ctx->activate();
        const uint64_t n = h->n_pairs;
        if (n == 0) return;
        const uint64_t nb = h->mate_off[2 * n], nr = h->aln_off[n], nc = counted ? 0 : h->cigar_off[n];
        if (h->mate_off[0] || h->aln_off[0] || (!counted && h->cigar_off[0])) fail(LCTY_ERR_INVALID_INPUT, "chunk offsets must start at 0");
        if (R->streaming && R->scored && R->n_pairs > R->raw_first) {
            // the chunk on the device has been scored: its products stay, its records make room for the next one
            R->check_device_error();
            LCTY_HIP(hipMemcpyAsync(&R->pa_at_raw_first, R->d_pa_count.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
            LCTY_HIP(hipStreamSynchronize(ctx->stream));
            R->raw_first = R->n_pairs;
            R->n_bases = R->n_recs = R->n_cigar = 0;
            // alignment recovery leaves record / CIGAR tables that are exactly as large as the merged chunk was
            if (R->d_recs.n < std::max<uint64_t>(R->chunk_cap_recs, 1)) R->d_recs.alloc(std::max<uint64_t>(R->chunk_cap_recs, 1));
            if (R->d_cigar.n < R->chunk_cap_cigar + 16) R->d_cigar.alloc(R->chunk_cap_cigar + 16);
            R->cap_recs = std::max<uint64_t>(R->chunk_cap_recs, R->d_recs.n); R->cap_cigar = R->d_cigar.n - 16;
        }
        const uint64_t raw_pairs = R->n_pairs - R->raw_first;       // pairs whose records are on the device
        if (R->n_pairs + n > R->cap_pairs || raw_pairs + n > R->cap_raw_pairs || R->n_bases + nb > R->cap_bases ||
            R->n_recs + nr > R->cap_recs || R->n_cigar + nc > R->cap_cigar)
            fail(LCTY_ERR_INVALID_INPUT, "chunk exceeds the capacity given to lcty_reads_create%s", R->streaming ? "_streaming" : "");
        // The bulk of the chunk is on its way before the host looks at it: from page-locked memory (lcty_host_alloc) the copies run while
        // the host validates; a chunk that fails validation leaves bytes behind the batch's counters, which nothing reads.
        // The copies run on the context's copy stream, not on the stream of its kernels: the kernels of another batch (the locus
        // before this one in a queue) do not hold them back, and this call waits for its own copies only. Whatever ends the call
        // early — a failed validation, an exception of the runtime — waits for the copies first: they read the caller's chunk.
        hipStream_t s = ctx->copy_stream();
        struct CopiesDone { hipStream_t s; bool armed = true; ~CopiesDone() { if (armed) (void)hipStreamSynchronize(s); } } copies_done{s};
        if (!dev && R->scored) LCTY_HIP(hipStreamSynchronize(ctx->stream));   // a scored batch that grows: its kernels have read the old tables
        static_assert(sizeof(lcty_aln_counted) == sizeof(lcty_aln_rec), "both record forms are 16 bytes");
        if (dev) {
            LCTY_HIP(hipStreamSynchronize(ctx->stream));                          // the mapping kernels that made the records
            if (R->n_bases / 16 + nb / 16 > R->d_bases2.n || R->n_recs + nr > R->d_recs.n || R->n_cigar + nc > R->d_cigar.n) fail(LCTY_ERR_RUNTIME, "device buffer overflow");
            LCTY_HIP(hipMemcpyAsync(R->d_bases2.p + R->n_bases / 16, dev->bases2, nb / 16 * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
            LCTY_HIP(hipMemcpyAsync(R->d_nmask.p + R->n_bases / 32, dev->nmask, nb / 32 * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
            if (nr) LCTY_HIP(hipMemcpyAsync(R->d_recs.p + R->n_recs, dev->recs, nr * sizeof(lcty_aln_rec), hipMemcpyDeviceToDevice, s));
            if (nc) LCTY_HIP(hipMemcpyAsync(R->d_cigar.p + R->n_cigar, dev->cigar, nc * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
        } else {
            R->d_recs.upload(counted ? reinterpret_cast<const lcty_aln_rec*>(counted) : h->recs, nr, s, R->n_recs);
            if (!counted) R->d_cigar.upload(h->cigar, nc, s, R->n_cigar);
            R->d_bases2.upload(h->bases2, nb / 16, s, R->n_bases / 16);
            R->d_nmask.upload(h->nmask, nb / 32, s, R->n_bases / 32);
        }
        // host-side validation of the CSR structure, O(pairs + records), on the host's cores (a chunk of 32 768 read pairs x 256 alleles is
        // 16 M records); the first offending pair decides the error, as a serial walk would
        uint32_t max_recs = R->max_recs_per_pair;
        uint64_t max_cig = R->max_cigar_per_pair;
        uint32_t max_rec_cig = R->max_cigar_per_rec;
        std::vector<uint2> meta(n);
        const bool paired = R->locus->bg.is_paired != 0;
        struct Problem { uint64_t pair = ~0ull; int32_t code = LCTY_OK; std::string text; };
        // lcty_ctx_set_knob "host_threads": the most host threads one call of the library starts (default 16; several ranks on one host share its cores)
        const uint64_t host_cap = static_cast<uint64_t>(std::max<int64_t>(1, ctx->knob("host_threads", 16)));
        const uint32_t n_threads = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>({nr / 500000 + 1, host_cap, std::thread::hardware_concurrency()})));
        std::vector<Problem> problems(n_threads);
        std::vector<uint32_t> t_max_recs(n_threads, 0), t_max_rec_cig(n_threads, 0);
        std::vector<uint64_t> t_max_cig(n_threads, 0);
        auto validate = [&](uint32_t tid) {
            const uint64_t r_lo = n * tid / n_threads, r_hi = n * (tid + 1) / n_threads;
            Problem& P = problems[tid];
            auto bad = [&](uint64_t r, int32_t code, const char* fmt, unsigned long long arg) {
                char buf[160];
                snprintf(buf, sizeof(buf), fmt, arg);
                P.pair = r; P.code = code; P.text = buf;
            };
            uint32_t mr = 0, mrc = 0; uint64_t mc = 0;
            for (uint64_t m = 2 * r_lo; m < 2 * r_hi; m++) {
                if (h->mate_off[m] % 32) { bad(m / 2, LCTY_ERR_INVALID_INPUT, "mate offsets must be multiples of 32 bases%.0llu", 0ull); return; }
                if (h->mate_off[m + 1] < h->mate_off[m] + h->mate_len[m]) { bad(m / 2, LCTY_ERR_INVALID_INPUT, "mate offsets overlap%.0llu", 0ull); return; }
            }
            for (uint64_t r = r_lo; r < r_hi; r++) {
                if (h->aln_off[r + 1] < h->aln_off[r] || (!counted && h->cigar_off[r + 1] < h->cigar_off[r])) {
                    bad(r, LCTY_ERR_INVALID_INPUT, "record / CIGAR offsets must be non-decreasing%.0llu", 0ull); return;
                }
                const uint64_t cnt = h->aln_off[r + 1] - h->aln_off[r];
                if (cnt > 0xFFFFFFFFull) { bad(r, LCTY_ERR_UNSUPPORTED, "too many records in one read pair%.0llu", 0ull); return; }
                mr = std::max<uint32_t>(mr, static_cast<uint32_t>(cnt));
                const uint64_t cw = counted ? 0 : h->cigar_off[r + 1] - h->cigar_off[r];
                mc = std::max(mc, cw);
                // record groups (locs.rs:1119-1131): the second primary starts read end 2, a third one would start
                // the next read pair
                uint32_t j2 = static_cast<uint32_t>(cnt), j3 = static_cast<uint32_t>(cnt);
                if (dev) {                                                        // one primary (or unmapped) record first for every read end that is there
                    if (dev->n_recs_mate[2 * r] + dev->n_recs_mate[2 * r + 1] != cnt) { bad(r, LCTY_ERR_RUNTIME, "record counts of the read ends do not add up%.0llu", 0ull); return; }
                    if (dev->n_recs_mate[2 * r + 1]) j2 = dev->n_recs_mate[2 * r];
                    mrc = std::max(mrc, dev->max_rec_cigar);
                }
                for (uint64_t i = h->aln_off[r]; !dev && i < h->aln_off[r + 1]; i++) {
                    bool is_primary;
                    if (counted) {
                        if (counted[i].pos_flags >> 31) { bad(r, LCTY_ERR_INVALID_INPUT, "counted alignment %llu: reserved flag bit set", (unsigned long long)i); return; }
                        is_primary = (counted[i].pos_flags & LCTY_CF_NOT_PRIMARY) == 0;
                    } else {
                        if (static_cast<uint64_t>(h->recs[i].cigar_rel) + h->recs[i].n_cigar > cw) {
                            bad(r, LCTY_ERR_INVALID_INPUT, "CIGAR of record %llu leaves its pair's CIGAR range", (unsigned long long)i); return;
                        }
                        mrc = std::max(mrc, h->recs[i].n_cigar);
                        is_primary = (h->recs[i].flags & (LCTY_FLAG_SECONDARY | LCTY_FLAG_SUPPL)) == 0;
                    }
                    const uint32_t idx = static_cast<uint32_t>(i - h->aln_off[r]);
                    if (idx > 0 && is_primary) {
                        if (j2 == cnt) j2 = idx; else if (j3 == cnt) j3 = idx;
                    }
                }
                meta[r] = make_uint2(j2, paired ? j3 : j2);
            }
            t_max_recs[tid] = mr; t_max_cig[tid] = mc; t_max_rec_cig[tid] = mrc;
        };
        {
            // threads that cannot be started: their share is walked here (a thread that did start is always joined)
            std::vector<std::thread> th;
            std::vector<uint32_t> here;
            for (uint32_t tid = 1; tid < n_threads; tid++) {
                try { th.emplace_back(validate, tid); }
                catch (const std::system_error&) { here.push_back(tid); }
            }
            validate(0);
            for (uint32_t tid : here) validate(tid);
            for (auto& x : th) x.join();
        }
        {
            const Problem* first = nullptr;
            for (const Problem& P : problems) if (P.code != LCTY_OK && (!first || P.pair < first->pair)) first = &P;
            if (first) {
                (void)hipStreamSynchronize(s);                                    // the copies read the caller's chunk
                fail(first->code, "%s", first->text.c_str());
            }
        }
        for (uint32_t tid = 0; tid < n_threads; tid++) {
            max_recs = std::max(max_recs, t_max_recs[tid]); max_cig = std::max(max_cig, t_max_cig[tid]); max_rec_cig = std::max(max_rec_cig, t_max_rec_cig[tid]);
        }
        try { R->locus->ensure_edit_thresholds(h->mate_len, 2 * n); }
        catch (...) { (void)hipStreamSynchronize(s); throw; }
        R->d_mate_len.upload(h->mate_len, 2 * n, s, 2 * raw_pairs);
        // rebased offsets
        std::vector<uint64_t> mo(2 * n), ao(n), co(n);
        for (uint64_t m = 0; m < 2 * n; m++) mo[m] = h->mate_off[m + 1] + R->n_bases;
        for (uint64_t r = 0; r < n; r++) { ao[r] = h->aln_off[r + 1] + R->n_recs; co[r] = (counted ? 0 : h->cigar_off[r + 1]) + R->n_cigar; }
        R->d_mate_off.upload(mo.data(), 2 * n, s, 2 * raw_pairs + 1);
        R->d_aln_off.upload(ao.data(), n, s, raw_pairs + 1);
        R->d_cigar_off.upload(co.data(), n, s, raw_pairs + 1);
        R->d_pair_meta.upload(meta.data(), n, s, raw_pairs);
        LCTY_HIP(hipStreamSynchronize(s));
        copies_done.armed = false;
        R->n_pairs += n; R->n_bases += nb; R->n_recs += nr; R->n_cigar += nc;
        R->max_recs_per_pair = max_recs;
        R->max_cigar_per_pair = static_cast<uint32_t>(std::min<uint64_t>(max_cig, 0xFFFFFFF0ull));
        R->max_cigar_per_rec = max_rec_cig;
        R->scored = false;
        R->counted = counted != nullptr;
        R->good_valid = false; R->loc_table_valid = false;
    });
}

int32_t lcty_reads_append(lcty_reads* R, const lcty_reads_host* h) { return append_impl(R, h, nullptr); }

extern "C++" {
namespace lcty {
int32_t reads_append_device(lcty_reads* R, const lcty_reads_host* h, const DeviceRecords* dev) { return append_impl(R, h, nullptr, dev); }
}
}

// the same chunk with its records already counted: h->recs / cigar_off / cigar are not read
int32_t lcty_reads_append_counted(lcty_reads* R, const lcty_reads_host* h, const lcty_aln_counted* alns) {
    if (!alns) return guarded([&] { fail(LCTY_ERR_INVALID_INPUT, "null argument"); });
    return append_impl(R, h, alns);
}



// An empty batch again, bound to `locus` (same context, no more alleles than the batch was made for): the buffers stay, so a queue of
// distinct loci rotates over a few batch objects instead of allocating tens of GB per locus (an allocation or a release waits
// for every stream of the device). The caller makes sure nothing of the batch is in use any more.
int32_t lcty_reads_reset(lcty_reads* R, lcty_locus* locus) {
    return guarded([&] {
        if (!R || !locus) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (locus->ctx != R->ctx) fail(LCTY_ERR_INVALID_INPUT, "the locus belongs to another context");
        if (static_cast<uint64_t>(locus->n_alleles) * R->cap_pairs > R->d_matrix.n)
            fail(LCTY_ERR_INVALID_INPUT, "the batch was made for a locus of fewer alleles (%u now)", locus->n_alleles);
        R->locus = locus;
        R->n_pairs = R->n_bases = R->n_recs = R->n_cigar = 0;
        R->max_recs_per_pair = R->max_cigar_per_pair = R->max_cigar_per_rec = 0;
        R->scored = false; R->counted = false; R->raw_first = 0; R->pa_at_raw_first = 0;
        R->good_valid = false; R->loc_table_valid = false; R->n_scores = 0; R->n_good_cached = 0;
        R->ctx->activate();
        hipStream_t s = R->ctx->copy_stream();
        R->d_err.zero(s); R->d_pa_count.zero(s);
        const uint64_t zero = 0;
        R->d_mate_off.upload(&zero, 1, s); R->d_aln_off.upload(&zero, 1, s);
        LCTY_HIP(hipStreamSynchronize(s));
    });
}

void lcty_reads_destroy(lcty_reads* reads) {
    if (!reads) return;
    (void)hipSetDevice(reads->ctx->device);
    delete reads;
}

int32_t lcty_reads_n_pairs(const lcty_reads* reads, uint64_t* out) {
    return guarded([&] {
        if (!reads || !out) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        *out = reads->n_pairs;
    });
}

int32_t lcty_score_reads(lcty_reads* reads) {
    return guarded([&] {
        if (!reads) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        reads->ctx->activate();
        if (reads->n_pairs > reads->raw_first) launch_score_reads(reads);      // streaming: the chunk on the device
        reads->scored = true;
        reads->good_valid = false; reads->loc_table_valid = false;
    });
}

static void require_scored(lcty_reads* reads) {
    if (!reads) fail(LCTY_ERR_INVALID_INPUT, "null argument");
    if (!reads->scored) fail(LCTY_ERR_INVALID_INPUT, "lcty_score_reads has not been called on this batch");
    reads->ctx->activate();
    reads->check_device_error();
}

int32_t lcty_reads_get_status(lcty_reads* reads, uint8_t* status, double* weight, double* unmapped_prob, uint16_t* uniq_kmers) {
    return guarded([&] {
        require_scored(reads);
        hipStream_t s = reads->ctx->stream;
        const uint64_t n = reads->n_pairs;
        if (status) reads->d_status.download(status, n, s);
        if (weight) reads->d_weight.download(weight, n, s);
        if (unmapped_prob) reads->d_unmapped.download(unmapped_prob, n, s);
        if (uniq_kmers) reads->d_uniq.download(uniq_kmers, 2 * n, s);
        LCTY_HIP(hipStreamSynchronize(s));
    });
}

int32_t lcty_reads_n_good(lcty_reads* reads, uint64_t* out) {
    return guarded([&] {
        require_scored(reads);
        if (!out) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        std::vector<uint8_t> status(reads->n_pairs);
        reads->d_status.download(status.data(), reads->n_pairs, reads->ctx->stream);
        LCTY_HIP(hipStreamSynchronize(reads->ctx->stream));
        uint64_t g = 0;
        for (uint8_t v : status) g += v == LCTY_READ_GOOD;
        *out = g;
    });
}

int32_t lcty_best_aln_matrix(lcty_reads* reads, double* out) {
    return guarded([&] {
        require_scored(reads);
        if (!out) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        uint64_t n_good = 0;
        {
            std::vector<uint8_t> status(reads->n_pairs);
            reads->d_status.download(status.data(), reads->n_pairs, reads->ctx->stream);
            LCTY_HIP(hipStreamSynchronize(reads->ctx->stream));
            for (uint8_t v : status) n_good += v == LCTY_READ_GOOD;
        }
        if (!n_good) return;
        const uint32_t A = reads->locus->n_alleles;
        DevBuf<double> d_out;
        d_out.alloc(n_good * A);
        compact_matrix(reads, d_out.p, n_good);
        d_out.download(out, n_good * A, reads->ctx->stream);
        LCTY_HIP(hipStreamSynchronize(reads->ctx->stream));
    });
}

// the record table as the batch holds it now — after lcty_recover_alignments the caller's records with the transferred alignments
// behind the records of their read end (what lcty_write_bam needs as its `table`)
int32_t lcty_reads_get_records(lcty_reads* reads, uint64_t* aln_off, lcty_aln_rec* recs, uint64_t cap_recs, uint64_t* cigar_off, uint32_t* cigar,
                               uint64_t cap_cigar) {
    return guarded([&] {
        if (!reads || !aln_off || !cigar_off) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (reads->counted) fail(LCTY_ERR_UNSUPPORTED, "a batch of counted alignments has no records");
        if (reads->streaming) fail(LCTY_ERR_UNSUPPORTED, "a streaming batch keeps the records of its current chunk only");
        reads->ctx->activate();
        hipStream_t s = reads->ctx->stream;
        const uint64_t n = reads->n_pairs;
        reads->d_aln_off.download(aln_off, n + 1, s);
        reads->d_cigar_off.download(cigar_off, n + 1, s);
        LCTY_HIP(hipStreamSynchronize(s));
        if (!recs && !cigar) return;                                        // sizing call: aln_off[n], cigar_off[n]
        if (!recs || !cigar) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (cap_recs < aln_off[n] || cap_cigar < cigar_off[n])
            fail(LCTY_ERR_INVALID_INPUT, "room for %llu records and %llu CIGAR words is needed", (unsigned long long)aln_off[n], (unsigned long long)cigar_off[n]);
        reads->d_recs.download(recs, aln_off[n], s);
        reads->d_cigar.download(cigar, cigar_off[n], s);
        LCTY_HIP(hipStreamSynchronize(s));
    });
}

int32_t lcty_reads_get_pair_alns(lcty_reads* reads, uint64_t* off, lcty_pair_aln* out, uint64_t cap) {
    return guarded([&] {
        require_scored(reads);
        hipStream_t s = reads->ctx->stream;
        const uint64_t n = reads->n_pairs;
        std::vector<uint64_t> dev_off(n);
        std::vector<uint32_t> cnt(n);
        reads->d_pa_off.download(dev_off.data(), n, s);
        reads->d_pa_cnt.download(cnt.data(), n, s);
        unsigned long long total = 0;
        LCTY_HIP(hipMemcpyAsync(&total, reads->d_pa_count.p, sizeof(total), hipMemcpyDeviceToHost, s));
        LCTY_HIP(hipStreamSynchronize(s));
        uint64_t run = 0;
        for (uint64_t r = 0; r < n; r++) { if (off) off[r] = run; run += cnt[r]; }
        if (off) off[n] = run;
        // the cursor counts what was reserved: equal to what is used unless wavefronts took the arena in chunks (lcty_score.hip)
        if (run > total) fail(LCTY_ERR_RUNTIME, "pair-alignment arena is inconsistent (%llu entries used, %llu reserved)", (unsigned long long)run, total);
        if (!out) return;
        if (cap < run) fail(LCTY_ERR_INVALID_INPUT, "output capacity %llu < %llu pair alignments", (unsigned long long)cap, (unsigned long long)run);
        std::vector<PairAlnDev> arena(total);
        reads->d_pa.download(arena.data(), total, s);
        LCTY_HIP(hipStreamSynchronize(s));
        uint64_t w = 0;
        for (uint64_t r = 0; r < n; r++) {
            for (uint32_t t = 0; t < cnt[r]; t++) {
                const PairAlnDev& d = arena[dev_off[r] + t];
                lcty_pair_aln& o = out[w++];
                memset(&o, 0, sizeof(o));
                o.ln_prob = d.ln_prob;
                o.ix1 = d.ix1 == 0xFFFFu ? LCTY_NONE_U32 : d.ix1;
                o.ix2 = d.ix2 == 0xFFFFu ? LCTY_NONE_U32 : d.ix2;
                o.mid1 = d.mid1; o.mid2 = d.mid2; o.contig = d.contig;
            }
        }
    });
}

}  // extern "C"
