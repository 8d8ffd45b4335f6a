// lcty_prefilter.hip — run_filter (src/solvers/solve.rs:87-122):
//     score[g] = prior[g] + sum_r max_{a in g} M[a][r]
//
// Diploid case (all multisets {i <= j}, the order of gen_combinations_with_repl, ext/vec.rs:298-339):
// a "max-plus Gram" reduction. The read-major matrix M[R][A] is streamed ONCE per 128x128 allele
// tile pair; a 256-thread workgroup keeps an 8x8 block of f64 accumulators per thread (all 16384
// genotype pairs of the tile) and walks 32-read stages staged through LDS. The read range is split
// over gridDim.y workgroups; partial sums land in a slab [split][G] and are added in split order
// by a second kernel, so the result is bitwise reproducible from run to run.
//
// f64 VALU-bound (2 ops per (genotype, read)); HBM traffic = A*R*8 B per tile-pair column.
#include "lcty_objects.hpp"

namespace lcty {

constexpr uint32_t PT = 128;        // alleles per tile side
constexpr uint32_t RC = 32;         // reads per LDS stage (and the unit the read range is split in)
constexpr uint32_t RC_BESIDE = 8;   // the same kernel with a quarter of the LDS (17 KB): the head of the NEXT locus of a queue runs beside the greedy
                                    // chains of the current one, whose two workgroups per CU leave 35 KB (lcty_solve_queue); same sums in the same order
constexpr uint32_t ROWD = PT + 8;   // doubles per LDS row: +16 B after every 32 doubles (bank spread for ds_read_b128)
constexpr uint32_t DIAG_BLOCKS = 136;   // 8x8 blocks (bi <= bj) of a diagonal tile: 16*17/2

__device__ inline uint32_t swz(uint32_t col) { return col + ((col >> 5) << 1); }

// max of two f64 as the single instruction it is: clang's fmax() quiets its operands first (one more v_max_f64 per
// loaded value under IEEE mode), which v_max_f64 already does by itself
__device__ __forceinline__ double vmax(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// genotype index of the multiset {i <= j} among all C(A+1, 2) (lexicographic; ext/vec.rs:298-339)
__host__ __device__ inline uint64_t gt_index(uint64_t i, uint64_t j, uint64_t A) { return i * A - i * (i - 1) / 2 + (j - i); }

template <uint32_t RCT>
__global__ __launch_bounds__(256) void prefilter_tile_kernel(const double* __restrict__ M, uint64_t n_reads, uint32_t A,
                                                             uint32_t n_tiles, uint64_t reads_per_split,
                                                             double* __restrict__ partials, uint64_t G) {
    extern __shared__ __align__(16) double lds[];
    double* lds_a = lds;
    double* lds_b = lds + RCT * ROWD;
    const uint32_t tid = threadIdx.x;
    // tile pair (I <= J) from the linear index
    uint32_t tp = blockIdx.x, I = 0;
    while (tp >= n_tiles - I) { tp -= n_tiles - I; I++; }
    const uint32_t J = I + tp;
    const bool diag = I == J;
    // 8x8 block of this thread
    uint32_t bi, bj;
    bool active = true;
    if (!diag) { bi = tid >> 4; bj = tid & 15u; }
    else {
        active = tid < DIAG_BLOCKS;
        uint32_t t = active ? tid : 0u; bi = 0;
        while (t >= 16u - bi) { t -= 16u - bi; bi++; }
        bj = bi + t;
    }
    double acc[8][8];
#pragma unroll
    for (int x = 0; x < 8; x++)
#pragma unroll
        for (int y = 0; y < 8; y++) acc[x][y] = 0.0;

    const uint64_t r_begin = static_cast<uint64_t>(blockIdx.y) * reads_per_split;
    const uint64_t r_end = min(n_reads, r_begin + reads_per_split);
    const uint32_t colI = I * PT, colJ = J * PT;
    if (diag) lds_b = lds_a;

    for (uint64_t r0 = r_begin; r0 < r_end; r0 += RCT) {
        // stage RC rows x 128 columns of both tiles (zero fill outside the matrix: max(0,0)+acc == acc); every load of
        // the stage is issued before the first LDS write
        constexpr uint32_t NT = RCT * PT / 256, HALF = NT / 2;
#pragma unroll
        for (uint32_t h = 0; h < 2; h++) {
            double va[HALF], vb[HALF];
#pragma unroll
            for (uint32_t t = 0; t < HALF; t++) {
                const uint32_t e = tid + 256u * (h * HALF + t);
                const uint32_t rr = e / PT, col = e % PT;
                const uint64_t r = r0 + rr;
                const bool rin = r < r_end;
                va[t] = (rin && colI + col < A) ? M[r * A + colI + col] : 0.0;
                vb[t] = (!diag && rin && colJ + col < A) ? M[r * A + colJ + col] : 0.0;
            }
#pragma unroll
            for (uint32_t t = 0; t < HALF; t++) {
                const uint32_t e = tid + 256u * (h * HALF + t);
                const uint32_t rr = e / PT, col = e % PT;
                lds_a[rr * ROWD + swz(col)] = va[t];
                if (!diag) lds_b[rr * ROWD + swz(col)] = vb[t];
            }
        }
        __syncthreads();
        if (active) {
            const double* pa = lds_a + swz(bi * 8);
            const double* pb = lds_b + swz(bj * 8);
#pragma unroll 2
            for (uint32_t rr = 0; rr < RCT; rr++) {
                double a[8], b[8];
#pragma unroll
                for (int x = 0; x < 8; x++) { a[x] = pa[rr * ROWD + x]; b[x] = pb[rr * ROWD + x]; }
#pragma unroll
                for (int x = 0; x < 8; x++)
#pragma unroll
                    for (int y = 0; y < 8; y++) acc[x][y] += vmax(a[x], b[y]);
            }
        }
        __syncthreads();
    }
    if (active) {
        double* out = partials + static_cast<uint64_t>(blockIdx.y) * G;
#pragma unroll
        for (int x = 0; x < 8; x++) {
            const uint64_t i = static_cast<uint64_t>(colI) + bi * 8 + x;
#pragma unroll
            for (int y = 0; y < 8; y++) {
                const uint64_t j = static_cast<uint64_t>(colJ) + bj * 8 + y;
                if (i < A && j < A && i <= j) out[gt_index(i, j, A)] = acc[x][y];
            }
        }
    }
}

// scores[g] = prior[g] + sum over splits, in split order
__global__ void prefilter_reduce_kernel(const double* __restrict__ partials, uint32_t n_splits, uint64_t G,
                                        const double* __restrict__ priors, double* __restrict__ scores) {
    const uint64_t g = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= G) return;
    double s = -0.0;
    for (uint32_t t = 0; t < n_splits; t++) s += partials[static_cast<uint64_t>(t) * G + g];
    scores[g] = (priors ? priors[g] : 0.0) + s;
}

// Any ploidy / explicit genotype list: one workgroup per genotype (general, not the fast path).
__global__ __launch_bounds__(256) void prefilter_generic_kernel(const double* __restrict__ M, uint64_t n_reads, uint32_t A,
                                                                const uint16_t* __restrict__ genotypes, uint32_t ploidy,
                                                                const double* __restrict__ priors, double* __restrict__ scores) {
    __shared__ double red[256];
    const uint64_t g = blockIdx.x;
    const uint16_t* ids = genotypes + g * ploidy;
    double local = 0.0;
    for (uint64_t r = threadIdx.x; r < n_reads; r += 256) {
        const double* row = M + r * A;
        double v = row[ids[0]];
        for (uint32_t t = 1; t < ploidy; t++) v = fmax(v, row[ids[t]]);
        local += v;
    }
    red[threadIdx.x] = local;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) scores[g] = (priors ? priors[g] : 0.0) + red[0];
}

uint64_t count_genotypes(uint32_t n_alleles, uint32_t ploidy) {
    // count_combinations(n + r - 1, r), src/ext/vec.rs:285-296
    const uint64_t n = static_cast<uint64_t>(n_alleles) + ploidy - 1, r = ploidy;
    if (r > n) return 0;
    const uint64_t m = std::min(r, n - r);
    uint64_t acc = 1;
    for (uint64_t v = 1; v <= m; v++) acc = acc * (n - v + 1) / v;
    return acc;
}

// the f64 tile kernel over R rows of a read-major matrix (the batch's own, or the rows lcty_gram.hip leaves to it)
void launch_prefilter_tile(lcty_reads* reads, const double* M, uint64_t R, double* d_scores_out) {
    lcty_ctx* ctx = reads->ctx;
    const uint32_t A = reads->locus->n_alleles;
    const uint64_t G = count_genotypes(A, 2);
    const uint32_t n_tiles = (A + PT - 1) / PT;
    const uint32_t n_tp = n_tiles * (n_tiles + 1) / 2;
    const uint32_t cus = static_cast<uint32_t>(ctx->props.multiProcessorCount);
    // enough workgroups for 2 per CU, but keep >= 8 stages per split so the slab stays small
    uint64_t splits = std::max<uint64_t>(1, (2ull * cus + n_tp - 1) / n_tp);
    const uint64_t max_splits = std::max<uint64_t>(1, (R + 8 * RC - 1) / (8 * RC));
    splits = std::min(splits, max_splits);
    uint64_t per = (R + splits - 1) / splits;
    per = (per + RC - 1) / RC * RC;
    if (per == 0) per = RC;
    splits = std::max<uint64_t>(1, (R + per - 1) / per);
    if (reads->d_partials.n < splits * G) reads->d_partials.alloc(splits * G);
    // the calling thread works on the context's fore stream: beside the greedy chains of another locus (RC_BESIDE)
    const bool beside = ctx->fore != nullptr && static_cast<hipStream_t>(ctx->stream) == ctx->fore;
    const size_t lds = 2 * (beside ? RC_BESIDE : RC) * ROWD * sizeof(double);
    // per device and cheap: no process-wide "already done" flag (contexts on several GPUs, several host threads)
    LCTY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(prefilter_tile_kernel<RC>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(2 * RC * ROWD * sizeof(double))));
    ctx->timed(LCTY_K_PREFILTER, [&] {
        if (beside)
            hipLaunchKernelGGL(prefilter_tile_kernel<RC_BESIDE>, dim3(n_tp, static_cast<uint32_t>(splits)), dim3(256), lds, ctx->stream,
                               M, R, A, n_tiles, per, reads->d_partials.p, G);
        else
            hipLaunchKernelGGL(prefilter_tile_kernel<RC>, dim3(n_tp, static_cast<uint32_t>(splits)), dim3(256), lds, ctx->stream,
                               M, R, A, n_tiles, per, reads->d_partials.p, G);
        hipLaunchKernelGGL(prefilter_reduce_kernel, dim3(static_cast<uint32_t>((G + 255) / 256)), dim3(256), 0, ctx->stream,
                           reads->d_partials.p, static_cast<uint32_t>(splits), G, static_cast<const double*>(nullptr),
                           d_scores_out);
    });
    LCTY_HIP(hipGetLastError());
}

void launch_prefilter_diploid(lcty_reads* reads) {
    if (launch_prefilter_gram(reads)) return;               // many alleles, few levels per row: the matrix cores (lcty_gram.hip)
    const uint64_t G = count_genotypes(reads->locus->n_alleles, 2);
    if (reads->d_scores.n < G) reads->d_scores.alloc(G);
    reads->n_scores = G;
    launch_prefilter_tile(reads, reads->d_matrix.p, reads->n_pairs, reads->d_scores.p);
}

void launch_prefilter_generic(lcty_reads* reads, const uint16_t* d_genotypes, uint64_t n_gt, uint32_t ploidy,
                              const double* d_priors, double* d_scores) {
    lcty_ctx* ctx = reads->ctx;
    if (n_gt == 0) return;
    if (n_gt > 0x7FFFFFFFull) fail(LCTY_ERR_UNSUPPORTED, "too many genotypes for the generic prefilter");
    ctx->timed(LCTY_K_PREFILTER, [&] {
        hipLaunchKernelGGL(prefilter_generic_kernel, dim3(static_cast<uint32_t>(n_gt)), dim3(256), 0, ctx->stream,
                           reads->d_matrix.p, reads->n_pairs, reads->locus->n_alleles, d_genotypes, ploidy, d_priors, d_scores);
    });
    LCTY_HIP(hipGetLastError());
}

}  // namespace lcty
