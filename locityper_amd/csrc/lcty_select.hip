// lcty_select.hip — truncate_ixs (src/solvers/solve.rs:52-84) on the scores run_filter left in HBM: at 4 096 alleles a locus has 8.4 M
// genotypes, and bringing 67 MB of scores to the host to select a few thousand of them cost more than the prefilter itself.
// The reference sorts ALL indices by (score descending, index ascending) and keeps a prefix; here the same order comes from one
// stable radix sort of (order-preserving 64-bit image of the score, index) pairs (rocPRIM through hipCUB), the prefix length from
// binary searches in the sorted keys (select_kernel, one lane), and only the kept indices travel.
#include <hipcub/hipcub.hpp>

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

__global__ void select_keys_kernel(const double* __restrict__ scores, uint64_t n, uint64_t* __restrict__ keys, uint64_t* __restrict__ ixs,
                                   uint32_t* __restrict__ nan_flag) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = scores[i];
    if (v != v) atomicOr(nan_flag, 1u);
    keys[i] = desc_key(v);
    ixs[i] = i;
}

// the length of the kept prefix, as truncate_ixs finds it (solve.rs:62-83): everything within filt_diff of the best; at least min_size
// (with everything tied with the min_size-th); at least `threads`
__global__ void select_kernel(const uint64_t* __restrict__ keys, uint64_t n, double filt_diff, uint64_t min_size, uint64_t threads,
                              unsigned long long* __restrict__ out) {
    if (threadIdx.x || blockIdx.x) return;
    // number of sorted entries with score >= t: the first position whose score is < t
    auto count_ge = [&](double t) -> uint64_t {
        uint64_t lo = 0, hi = n;
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo) / 2;
            if (key_score(keys[mid]) >= t) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    const double best = key_score(keys[0]), worst = key_score(keys[n - 1]);
    double thresh = best - filt_diff;
    uint64_t m;
    if (min_size >= n || worst >= thresh) m = n;
    else {
        m = count_ge(thresh);
        if (m < min_size) { thresh = key_score(keys[min_size - 1]); m = count_ge(thresh); }
        m = m > threads ? m : threads;
        m = m < n ? m : n;
    }
    out[0] = m;
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
        lcty_ctx* ctx = reads->ctx;
        ctx->activate();
        reads->check_device_error();
        hipStream_t s = ctx->stream;
        auto& B = reads->select;
        B.k_in.ensure(n); B.k_out.ensure(n); B.v_in.ensure(n); B.v_out.ensure(n); B.out.ensure(2);
        LCTY_HIP(hipMemsetAsync(B.out.p, 0, 2 * sizeof(unsigned long long), s));
        uint32_t* flag = reinterpret_cast<uint32_t*>(B.out.p + 1);
        hipLaunchKernelGGL(select_keys_kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, s, reads->d_scores.p, n, B.k_in.p, B.v_in.p, flag);
        LCTY_HIP(hipGetLastError());
        size_t tmp_bytes = 0;
        LCTY_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, B.k_in.p, B.k_out.p, B.v_in.p, B.v_out.p, static_cast<uint64_t>(n), 0, 64, s));
        B.tmp.ensure(tmp_bytes ? tmp_bytes : 1);
        LCTY_HIP(hipcub::DeviceRadixSort::SortPairs(B.tmp.p, tmp_bytes, B.k_in.p, B.k_out.p, B.v_in.p, B.v_out.p, static_cast<uint64_t>(n), 0, 64, s));   // stable: ties stay in index order
        hipLaunchKernelGGL(select_kernel, dim3(1), dim3(64), 0, s, B.k_out.p, n, filt_diff, min_size, threads, B.out.p);
        LCTY_HIP(hipGetLastError());
        unsigned long long res[2] = {0, 0};
        B.out.download(res, 2, s);
        LCTY_HIP(hipStreamSynchronize(s));
        const unsigned long long m = res[0];
        const uint32_t nan = static_cast<uint32_t>(res[1]);
        // the reference never sees a NaN here (lcty_truncate's comment): a caller's own priors are refused instead of ordered somehow
        if (nan) fail(LCTY_ERR_INVALID_INPUT, "a prefilter score is NaN");
        *n_keep = m;
        if (!ixs) return;                                                  // sizing call
        if (m > cap) fail(LCTY_ERR_INVALID_INPUT, "room for %llu kept genotypes is needed", m);
        B.v_out.download(ixs, m, s);
        LCTY_HIP(hipStreamSynchronize(s));
    });
}

}  // extern "C"
