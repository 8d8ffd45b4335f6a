// lcty_select.hip — truncate_ixs (src/solvers/solve.rs:52-84) on the scores run_filter left in HBM: at 4 096 alleles a locus has 8.4 M
// genotypes, and bringing 67 MB of scores to the host to select a few thousand of them cost more than the prefilter itself.
// The reference sorts ALL indices by (score descending, index ascending) and keeps a prefix. Only the prefix is needed, so nothing
// is sorted that is not kept:
//   keys      order-preserving 64-bit image of every score (ascending key = descending score), smallest and largest key (atomics);
//   decide    the threshold `best - filt_diff` as a key, or "keep everything" (solve.rs:66-69);
//   count     keys within the threshold (partition_point, solve.rs:72);
//   select    when that is fewer than min_size (or than `threads`): the key at that rank by radix selection — eight passes of a
//             256-bin histogram over the keys that share the prefix found so far, the pick made by the pass's last workgroup —
//             and the count again (everything tied with that score stays, solve.rs:73-76; at least `threads`, 79);
//   compact   (key, index) of every genotype within the final key, in any order (a cursor per wavefront);
//   sort      the survivors by (key, index) in the LDS of one workgroup (bitonic, up to 8 192 of them; the default scheme keeps
//             5 000); more than that are ordered by the host from the compacted pairs.
// Only the kept indices travel. Equal to the host form lcty_truncate index for index (tests/helpers check_prefilter).
#include "lcty_objects.hpp"

namespace lcty {

namespace {

// f64 -> u64 whose ASCENDING order is the scores' DESCENDING order (total order on everything but NaN; -0.0 sorts behind +0.0 as
// f64::total_cmp has it, solve.rs:60)
__host__ __device__ inline uint64_t desc_key(double v) {
    uint64_t b;
    memcpy(&b, &v, 8);
    const uint64_t asc = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    return ~asc;
}
__host__ __device__ inline double key_score(uint64_t k) {
    const uint64_t asc = ~k;
    const uint64_t b = (asc >> 63) ? (asc & 0x7FFFFFFFFFFFFFFFull) : ~asc;
    double v;
    memcpy(&v, &b, 8);
    return v;
}

// control block of one lcty_prefilter_truncate (device memory, 64-bit words)
enum SelWord : uint32_t {
    SEL_MIN_KEY = 0, SEL_MAX_KEY, SEL_NAN, SEL_MODE_ALL, SEL_BOUND /* keep keys <= this */, SEL_M /* length of the kept prefix */,
    SEL_COUNT /* keys <= bound */, SEL_K /* rank the running selection looks for (0: none) */, SEL_PREFIX, SEL_TICKET, SEL_CAND /* compacted pairs */,
    SEL_RECOUNT /* the next sel_count_kernel counts */, SEL_DONE /* SEL_M is final */,
    SEL_WORDS = 16, SEL_HIST = 16 /* 256 bins behind the words */
};
constexpr uint32_t SORT_LDS_MAX = 8192;

__global__ __launch_bounds__(256) void sel_keys_kernel(const double* __restrict__ scores, uint64_t n, uint64_t* __restrict__ keys,
                                                       unsigned long long* __restrict__ ctl) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
    uint64_t k_min = ~0ull, k_max = 0ull;
    bool nan = false;
    if (i < n) {
        const double v = scores[i];
        nan = v != v;
        const uint64_t k = desc_key(v);
        keys[i] = k;
        k_min = k; k_max = k;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t a = __shfl_xor(k_min, o), b = __shfl_xor(k_max, o);
        k_min = a < k_min ? a : k_min; k_max = b > k_max ? b : k_max;
    }
    if ((threadIdx.x & 63u) == 0) { atomicMin(&ctl[SEL_MIN_KEY], static_cast<unsigned long long>(k_min)); atomicMax(&ctl[SEL_MAX_KEY], static_cast<unsigned long long>(k_max)); }
    if (__ballot(nan) && (threadIdx.x & 63u) == 0) atomicOr(&ctl[SEL_NAN], 1ull);
}

// solve.rs:64-69: the threshold, or everything
__global__ void sel_decide_kernel(unsigned long long* __restrict__ ctl, uint64_t n, double filt_diff, uint64_t min_size) {
    if (threadIdx.x || blockIdx.x) return;
    const double best = key_score(ctl[SEL_MIN_KEY]), worst = key_score(ctl[SEL_MAX_KEY]);
    double thresh = best - filt_diff;
    if (min_size >= n || worst >= thresh) { ctl[SEL_MODE_ALL] = 1; ctl[SEL_BOUND] = ~0ull; ctl[SEL_M] = n; ctl[SEL_DONE] = 1; return; }
    if (thresh == 0.0) thresh = -0.0;                       // `score >= thresh` holds for both zeros: the bound is the later of the two keys
    ctl[SEL_BOUND] = desc_key(thresh);
    ctl[SEL_RECOUNT] = 1;
}

// keys <= bound -> SEL_COUNT (zeroed by the caller)
__global__ __launch_bounds__(256) void sel_count_kernel(const uint64_t* __restrict__ keys, uint64_t n, unsigned long long* __restrict__ ctl) {
    if (ctl[SEL_MODE_ALL] || !ctl[SEL_RECOUNT]) return;
    const uint64_t bound = ctl[SEL_BOUND];
    uint32_t mine = 0;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<uint64_t>(gridDim.x) * 256) mine += keys[i] <= bound;
    for (int o = 32; o > 0; o >>= 1) mine += static_cast<uint32_t>(__shfl_xor(static_cast<int>(mine), o));
    if ((threadIdx.x & 63u) == 0 && mine) atomicAdd(&ctl[SEL_COUNT], static_cast<unsigned long long>(mine));
}

// what the count means (solve.rs:72-80). phase 0: after the count within the threshold; phase 1: after the count within the
// min_size-th score, if that was looked up. Sets SEL_K when a rank has to be looked up next (and clears the count for the recount
// behind it); SEL_M once it is known: everything within the bound — or exactly `threads` when fewer than that are within it (the
// list is cut there, ties or not: ixs.truncate(m)).
__global__ void sel_rank_kernel(unsigned long long* __restrict__ ctl, uint64_t n, uint64_t min_size, uint64_t threads, uint32_t phase) {
    if (threadIdx.x || blockIdx.x) return;
    ctl[SEL_K] = 0; ctl[SEL_RECOUNT] = 0;
    if (ctl[SEL_MODE_ALL] || ctl[SEL_DONE]) return;
    const uint64_t c = ctl[SEL_COUNT];
    const uint64_t at_least = threads < n ? threads : n;
    if (phase == 0 && c < min_size) {                        // the min_size-th score becomes the threshold (solve.rs:73-76)
        ctl[SEL_K] = min_size; ctl[SEL_PREFIX] = 0; ctl[SEL_COUNT] = 0; ctl[SEL_RECOUNT] = 1;
        return;
    }
    ctl[SEL_DONE] = 1;
    if (c >= at_least) { ctl[SEL_M] = c; return; }
    ctl[SEL_M] = at_least;                                    // solve.rs:79
    ctl[SEL_K] = at_least; ctl[SEL_PREFIX] = 0; ctl[SEL_COUNT] = 0; ctl[SEL_RECOUNT] = 1;
}

// one pass of the radix selection of the SEL_K-th smallest key: histogram of byte `pass` (from the top) over the keys whose
// higher bytes equal the prefix; the last workgroup to finish picks the bin and extends the prefix. After pass 7 the prefix is
// the key: it becomes the bound.
__global__ __launch_bounds__(256) void sel_radix_kernel(const uint64_t* __restrict__ keys, uint64_t n, unsigned long long* __restrict__ ctl, uint32_t pass) {
    __shared__ uint32_t hist[256];
    __shared__ bool last;
    if (ctl[SEL_K] == 0) return;
    hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t shift = 56 - 8 * pass;
    const uint64_t prefix = ctl[SEL_PREFIX];
    const uint64_t high_mask = pass == 0 ? 0ull : ~0ull << (shift + 8);
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<uint64_t>(gridDim.x) * 256) {
        const uint64_t k = keys[i];
        if ((k & high_mask) == prefix) atomicAdd(&hist[(k >> shift) & 0xFFu], 1u);
    }
    __syncthreads();
    unsigned long long* ghist = ctl + SEL_HIST;
    if (hist[threadIdx.x]) atomicAdd(&ghist[threadIdx.x], static_cast<unsigned long long>(hist[threadIdx.x]));
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(&ctl[SEL_TICKET], 1ull) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    if (threadIdx.x == 0) {
        uint64_t k = ctl[SEL_K], before = 0;
        uint32_t b = 0;
        for (; b < 256; b++) {
            const uint64_t h = __hip_atomic_load(&ghist[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (before + h >= k) break;
            before += h;
        }
        ctl[SEL_K] = k - before;
        const uint64_t p = prefix | (static_cast<uint64_t>(b) << shift);
        ctl[SEL_PREFIX] = p;
        ctl[SEL_TICKET] = 0;
        if (pass == 7) ctl[SEL_BOUND] = p;
    }
    __syncthreads();
    ghist[threadIdx.x] = 0;
}

// (key, index) of every genotype within the bound, in any order
__global__ __launch_bounds__(256) void sel_compact_kernel(const uint64_t* __restrict__ keys, uint64_t n, unsigned long long* __restrict__ ctl,
                                                          uint64_t* __restrict__ cand_key, uint64_t* __restrict__ cand_ix, uint64_t cap) {
    const uint64_t bound = ctl[SEL_BOUND];
    for (uint64_t base = static_cast<uint64_t>(blockIdx.x) * 256; base < n; base += static_cast<uint64_t>(gridDim.x) * 256) {
        const uint64_t i = base + threadIdx.x;
        const uint64_t k = i < n ? keys[i] : ~0ull;
        const bool in = i < n && k <= bound;
        const unsigned long long mask = __ballot(in);
        if (!mask) continue;
        const uint32_t lane = threadIdx.x & 63u;
        unsigned long long at = 0;
        if (lane == 0) at = atomicAdd(&ctl[SEL_CAND], static_cast<unsigned long long>(__popcll(mask)));
        at = __shfl(at, 0) + static_cast<unsigned long long>(__popcll(mask & ((1ull << lane) - 1ull)));
        if (in && at < cap) { cand_key[at] = k; cand_ix[at] = i; }
    }
}

// the survivors by (key, index): bitonic sort of up to SORT_LDS_MAX pairs in the LDS of one workgroup; the first SEL_M indices out
__global__ __launch_bounds__(1024) void sel_sort_kernel(const uint64_t* __restrict__ cand_key, const uint64_t* __restrict__ cand_ix,
                                                        unsigned long long* __restrict__ ctl, uint64_t* __restrict__ out_ix) {
    extern __shared__ __align__(16) uint8_t smem[];
    const uint64_t c = ctl[SEL_CAND];
    if (c > SORT_LDS_MAX) return;                           // the host orders them
    uint32_t N = 1;
    while (N < c) N <<= 1;
    uint64_t* key = reinterpret_cast<uint64_t*>(smem);
    uint32_t* ix = reinterpret_cast<uint32_t*>(key + N);
    for (uint32_t i = threadIdx.x; i < N; i += 1024) { key[i] = i < c ? cand_key[i] : ~0ull; ix[i] = i < c ? static_cast<uint32_t>(cand_ix[i]) : 0xFFFFFFFFu; }
    __syncthreads();
    for (uint32_t size = 2; size <= N; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = threadIdx.x; t < N / 2; t += 1024) {
                const uint32_t lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint64_t ka = key[lo], kb = key[hi];
                const uint32_t ia = ix[lo], ib = ix[hi];
                const bool a_after_b = ka > kb || (ka == kb && ia > ib);
                if (a_after_b == up) { key[lo] = kb; key[hi] = ka; ix[lo] = ib; ix[hi] = ia; }
            }
            __syncthreads();
        }
    }
    const uint64_t m = ctl[SEL_M];
    for (uint32_t i = threadIdx.x; i < m && i < c; i += 1024) out_ix[i] = ix[i];
}

__global__ void add_priors_kernel(double* __restrict__ scores, const double* __restrict__ priors, uint64_t n) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) scores[i] = priors[i] + scores[i];                                // prior + sum (solve.rs:114)
}

}  // namespace

}  // namespace lcty

using namespace lcty;

extern "C" {

// scores[g] = priors[g] + scores[g] on the device (run_filter adds the prior of a genotype to its sum, solve.rs:114): the step between
// lcty_prefilter_async (/ lcty_prefilter_allreduce) and lcty_prefilter_truncate when `--priors` is given
int32_t lcty_prefilter_add_priors(lcty_reads* reads, const double* priors, uint64_t n) {
    return guarded([&] {
        if (!reads || !priors) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (!reads->scored || reads->n_scores == 0 || reads->d_scores.n < reads->n_scores)
            fail(LCTY_ERR_INVALID_INPUT, "no prefilter scores on the device: lcty_prefilter_async first");
        if (n != reads->n_scores) fail(LCTY_ERR_INVALID_INPUT, "expected %llu priors", static_cast<unsigned long long>(reads->n_scores));
        lcty_ctx* ctx = reads->ctx;
        ctx->activate();
        hipStream_t s = ctx->stream;
        auto& B = reads->select;
        B.k_in.ensure(n);                                                        // staging for the priors (the sort's key buffer: free until the sort)
        LCTY_HIP(hipMemcpyAsync(B.k_in.p, priors, n * sizeof(double), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(add_priors_kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, s, reads->d_scores.p,
                           reinterpret_cast<const double*>(B.k_in.p), n);
        LCTY_HIP(hipGetLastError());
        LCTY_HIP(hipStreamSynchronize(s));                                       // `priors` is the caller's memory
    });
}

int32_t lcty_prefilter_truncate(lcty_reads* reads, double filt_diff, uint64_t min_size, uint64_t threads, uint64_t* ixs, uint64_t cap,
                                uint64_t* n_keep) {
    return guarded([&] {
        if (!reads || !n_keep) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (!reads->scored) fail(LCTY_ERR_INVALID_INPUT, "lcty_score_reads and a prefilter call first");
        const uint64_t n = reads->n_scores;
        if (n == 0 || reads->d_scores.n < n) fail(LCTY_ERR_INVALID_INPUT, "no prefilter scores on the device: lcty_prefilter_async (or lcty_prefilter) first");
        if (n >= (1ull << 32)) fail(LCTY_ERR_UNSUPPORTED, "2^32 or more genotypes: lcty_prefilter_scores + lcty_truncate");
        lcty_ctx* ctx = reads->ctx;
        ctx->activate();
        reads->check_device_error();
        hipStream_t s = ctx->stream;
        auto& B = reads->select;
        // k_in: keys; k_out / v_out: compacted pairs; v_in: sorted indices; out: the control block
        B.k_in.ensure(n); B.k_out.ensure(n); B.v_out.ensure(n); B.v_in.ensure(std::max<uint64_t>(SORT_LDS_MAX, 1)); B.out.ensure(SEL_WORDS + 256);
        unsigned long long* ctl = B.out.p;
        LCTY_HIP(hipMemsetAsync(ctl, 0, (SEL_WORDS + 256) * sizeof(unsigned long long), s));
        LCTY_HIP(hipMemsetAsync(ctl + SEL_MIN_KEY, 0xFF, sizeof(unsigned long long), s));
        const uint32_t blocks = static_cast<uint32_t>((n + 255) / 256), sweep = std::min<uint32_t>(blocks, 2048);
        hipLaunchKernelGGL(sel_keys_kernel, dim3(blocks), dim3(256), 0, s, reads->d_scores.p, n, B.k_in.p, ctl);
        hipLaunchKernelGGL(sel_decide_kernel, dim3(1), dim3(64), 0, s, ctl, n, filt_diff, min_size);
        hipLaunchKernelGGL(sel_count_kernel, dim3(sweep), dim3(256), 0, s, B.k_in.p, n, ctl);
        for (uint32_t phase = 0; phase < 2; phase++) {
            hipLaunchKernelGGL(sel_rank_kernel, dim3(1), dim3(64), 0, s, ctl, n, min_size, threads, phase);
            for (uint32_t pass = 0; pass < 8; pass++) hipLaunchKernelGGL(sel_radix_kernel, dim3(sweep), dim3(256), 0, s, B.k_in.p, n, ctl, pass);
            hipLaunchKernelGGL(sel_count_kernel, dim3(sweep), dim3(256), 0, s, B.k_in.p, n, ctl);     // counts only behind a selection (SEL_RECOUNT): else the count stands
        }
        hipLaunchKernelGGL(sel_compact_kernel, dim3(sweep), dim3(256), 0, s, B.k_in.p, n, ctl, B.k_out.p, B.v_out.p, n);
        const size_t sort_lds = static_cast<size_t>(SORT_LDS_MAX) * 12;
        LCTY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sel_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(sort_lds)));
        hipLaunchKernelGGL(sel_sort_kernel, dim3(1), dim3(1024), sort_lds, s, B.k_out.p, B.v_out.p, ctl, B.v_in.p);
        LCTY_HIP(hipGetLastError());
        unsigned long long res[SEL_WORDS];
        B.out.download(res, SEL_WORDS, s);
        LCTY_HIP(hipStreamSynchronize(s));
        // the reference never sees a NaN here (lcty_truncate's comment): a caller's own priors are refused instead of ordered somehow
        if (res[SEL_NAN]) fail(LCTY_ERR_INVALID_INPUT, "a prefilter score is NaN");
        const uint64_t m = res[SEL_M], c = res[SEL_CAND];
        if (m > c || c > n) fail(LCTY_ERR_RUNTIME, "truncate_ixs on the device lost count (%llu kept of %llu within the bound)", (unsigned long long)m, (unsigned long long)c);
        *n_keep = m;
        if (!ixs) return;                                                  // sizing call
        if (m > cap) fail(LCTY_ERR_INVALID_INPUT, "room for %llu kept genotypes is needed", (unsigned long long)m);
        if (c <= SORT_LDS_MAX) B.v_in.download(ixs, m, s);
        else {
            // more survivors than one workgroup orders: the compacted pairs come over and the host orders them
            std::vector<uint64_t> k(c), v(c);
            B.k_out.download(k.data(), c, s); B.v_out.download(v.data(), c, s);
            LCTY_HIP(hipStreamSynchronize(s));
            std::vector<uint64_t> order(c);
            for (uint64_t i = 0; i < c; i++) order[i] = i;
            std::sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) { return k[a] != k[b] ? k[a] < k[b] : v[a] < v[b]; });
            for (uint64_t i = 0; i < m; i++) ixs[i] = v[order[i]];
        }
        LCTY_HIP(hipStreamSynchronize(s));
    });
}

}  // extern "C"
