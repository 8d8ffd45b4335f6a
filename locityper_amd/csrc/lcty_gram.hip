// lcty_gram.hip — run_filter (src/solvers/solve.rs:87-122) of a diploid locus with many alleles as an EXACT integer Gram
// contraction on the matrix cores (BASELINE configs[4]: 4 096 alleles).
//
//     score[{i, j}] = sum over reads r of max(M[r][i], M[r][j])
//
// A row of the likelihood matrix takes few distinct values across the alleles (4.5 on average at 256 alleles, 5.1 at 1 024:
// an allele either carries the read's best location, or one of a few worse ones, or none). With the row's distinct values
// v_1 > v_2 > ... > v_L and N_k[a] = [M[r][a] < v_k]  ("allele a is below level k"):
//
//     max(M[r][i], M[r][j]) = v_1 - sum_{k=1}^{L-1} (v_k - v_{k+1}) * N_k[i] * N_k[j]
//
// so   score = C - sum over columns c = (r, k) of delta_c * N[c][i] * N[c][j],   C = sum_r v_1(r):
// a Gram matrix of 0/1 columns with non-negative weights. The weights become 35-bit fixed point (scaled by a power of two so
// that the largest fills the 35 bits), cut into five 7-bit digit planes; per plane the contraction is
//     S_p[i][j] = sum_c (digit_p[c] * N[c][i]) * N[c][j]        i8 x i8 -> i32, v_mfma_i32_32x32x32_i8
// and S = sum_p S_p << 7p is an exact 64-bit integer whatever the order of the columns — run to run reproducible, and within
// 2^-36 of the largest level difference per column of the f64 sum the reference makes (the tests hold it to 1e-9 relative,
// as the f64 tile kernel of lcty_prefilter.hip).
//
//   gram_levels_kernel    per block of 128 reads: the levels of every row (a wavefront holds a row in registers), the columns'
//                         weights, and the 0/1 columns as BITS, allele-major: word [a][c / 32], so that a lane of the
//                         contraction reads the 32 columns of a k-step of "its" allele with one load. Column ranges of a block
//                         are padded to whole words with weight-0 columns. Rows with more than 16 levels (or non-finite entries)
//                         are left to the f64 kernel ("residual rows").
//   gram_digits_kernel    the scale (from the largest weight) and the digit planes.
//   gram_mfma_kernel      128 x 128 alleles per workgroup of eight wavefronts, 64 x 32 per wavefront (two tiles of 32 x 32, five
//                         planes each: 160 accumulator registers); operands are made in registers from the bits (a nibble spread to four
//                         bytes by one multiplication) and the digits; both operands take their k order from the same bits, so
//                         the sum over k is right whatever the instruction's internal k order is.
//   gram_finish_kernel    score = C - S * 2^-F (+ the residual rows' f64 partial).
#include "lcty_objects.hpp"

namespace lcty {

namespace {

constexpr uint32_t GR_LMAX = 16;       // levels of a row the Gram form takes
constexpr uint32_t GR_RB = 128;        // rows per workgroup of the level kernel
constexpr uint32_t GR_PLANES = 5;      // 7-bit digit planes
constexpr uint32_t GR_BITS = 7 * GR_PLANES;

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ double wave_max_f64(double x) {
    for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o));
    return x;
}

// counters[0] columns handed out, [1] overflow of the column capacity, [2] residual rows, [3] largest weight (bits of the f64)
template <uint32_t VPL>
__global__ __launch_bounds__(256) void gram_levels_kernel(const double* __restrict__ M, uint64_t n_rows, uint32_t A, uint64_t k_cap, uint64_t kw,
                                                          uint32_t* __restrict__ bits, double* __restrict__ delta, double* __restrict__ c_part,
                                                          uint8_t* __restrict__ residual, unsigned long long* __restrict__ counters, uint32_t lmax) {
    __shared__ double lv[GR_RB][GR_LMAX];
    __shared__ uint32_t ncol[GR_RB], off[GR_RB];
    __shared__ double c_row[GR_RB];
    __shared__ unsigned long long base_sh;
    __shared__ uint32_t total_sh;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t r0 = static_cast<uint64_t>(blockIdx.x) * GR_RB;
    // ---- the levels of every row of the block: a wavefront holds a row in registers
    for (uint32_t rr = wave; rr < GR_RB; rr += 4) {
        const uint64_t r = r0 + rr;
        uint32_t L = 0;
        bool bad = false;
        double first = 0.0;
        if (r < n_rows) {
            double x[VPL];
#pragma unroll
            for (uint32_t t = 0; t < VPL; t++) {
                const uint32_t a = t * 64 + lane;
                x[t] = a < A ? M[r * A + a] : -INFINITY;
                bad |= a < A && !(fabs(x[t]) <= 1.7e308);                      // infinities and NaN: the reference's own f64 sum decides
            }
            bad = __any(bad);
            double prev = INFINITY;
            for (uint32_t k = 0; k <= lmax && !bad; k++) {
                double cand = -INFINITY;
#pragma unroll
                for (uint32_t t = 0; t < VPL; t++) cand = fmax(cand, x[t] < prev ? x[t] : -INFINITY);
                cand = wave_max_f64(cand);
                if (cand == -INFINITY) break;
                if (k == lmax) { bad = true; break; }                          // one level too many (lmax <= 16)
                if (lane == 0) lv[rr][k] = cand;
                if (k == 0) first = cand;
                prev = cand; L = k + 1;
            }
        }
        if (lane == 0) {
            ncol[rr] = (bad || L == 0) ? 0u : L - 1;
            c_row[rr] = (bad || L == 0) ? 0.0 : first;
            if (r < n_rows) residual[r] = bad ? 1 : 0;
        }
    }
    __syncthreads();
    // ---- columns of the block: prefix over its rows, a whole number of 32-column words from the global counter
    if (wave == 0) {
        const uint32_t n0 = ncol[2 * lane], n1 = ncol[2 * lane + 1];
        uint32_t incl = n0 + n1;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(incl, o); if (lane >= static_cast<uint32_t>(o)) incl += up; }
        off[2 * lane] = incl - n0 - n1; off[2 * lane + 1] = incl - n1;
        // sum of the rows' largest values, rows in order (reproducible)
        double c = 0.0;
        if (lane == 0) { for (uint32_t i = 0; i < GR_RB; i++) c += c_row[i]; c_part[blockIdx.x] = c; }
        if (lane == 63) {
            const uint32_t padded = (incl + 31u) & ~31u;
            total_sh = incl;
            unsigned long long b = 0;
            if (padded) b = atomicAdd(&counters[0], static_cast<unsigned long long>(padded));
            if (b + padded > k_cap) { atomicMax(&counters[1], 1ull); b = ~0ull; }
            base_sh = b;
        }
    }
    __syncthreads();
    const unsigned long long base = base_sh;
    if (base == ~0ull || total_sh == 0) return;                                 // out of room (the caller falls back), or nothing to write
    // weights of the columns (the padding columns keep the 0 the buffer was cleared to)
    for (uint32_t rr = tid; rr < GR_RB; rr += 256) {
        double big = 0.0;
        for (uint32_t k = 0; k < ncol[rr]; k++) {
            const double d = lv[rr][k] - lv[rr][k + 1];
            delta[base + off[rr] + k] = d;
            big = fmax(big, d);
        }
        if (big > 0.0) atomicMax(&counters[3], static_cast<unsigned long long>(__double_as_longlong(big)));
    }
    // ---- the 0/1 columns as bits, allele-major: this thread's alleles, the block's rows in order
    const uint64_t w0 = base / 32;
    for (uint32_t a = tid; a < A; a += 256) {
        uint32_t word = 0, nb = 0;
        uint64_t w = w0;
        for (uint32_t r4 = 0; r4 < GR_RB; r4 += 4) {
            // four rows' cells in flight (rows without columns — and rows behind the end of the matrix — are not read)
            double x[4];
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) x[q] = ncol[r4 + q] ? M[(r0 + r4 + q) * A + a] : 0.0;
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) {
                const uint32_t rr = r4 + q, n = ncol[rr];
                for (uint32_t k = 0; k < n; k++) {
                    word |= (x[q] < lv[rr][k] ? 1u : 0u) << nb;
                    if (++nb == 32) { bits[static_cast<uint64_t>(a) * kw + w++] = word; word = 0; nb = 0; }
                }
            }
        }
        if (nb) bits[static_cast<uint64_t>(a) * kw + w] = word;
    }
}

// scale = 2^F with the largest weight in [2^34, 2^35); digit p of column c = bits 7p .. 7p + 6 of round(weight * scale)
__global__ __launch_bounds__(256) void gram_digits_kernel(const double* __restrict__ delta, uint64_t n_cols, uint64_t k_cap,
                                                          const unsigned long long* __restrict__ counters, uint8_t* __restrict__ dig,
                                                          int* __restrict__ f_out) {
    const double dmax = __longlong_as_double(static_cast<long long>(counters[3]));
    const int F = dmax > 0.0 ? static_cast<int>(GR_BITS) - 1 - ilogb(dmax) : 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) *f_out = F;
    const uint64_t c = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (c >= n_cols) return;
    unsigned long long q = static_cast<unsigned long long>(llrint(ldexp(delta[c], F)));
    q = min(q, (1ull << GR_BITS) - 1ull);
#pragma unroll
    for (uint32_t p = 0; p < GR_PLANES; p++) dig[static_cast<uint64_t>(p) * k_cap + c] = static_cast<uint8_t>((q >> (7 * p)) & 127ull);
}

// 16 bits -> 16 bytes of 0 / 1 (four dwords): a nibble times 0x00204081 puts bit i at bit 8i (a 256-entry table in LDS was
// slower: bank conflicts and a dependent LDS round trip per k-step)
__device__ __forceinline__ v4i spread16(uint32_t b) {
    v4i e;
    e.x = static_cast<int>(((b & 0xFu) * 0x00204081u) & 0x01010101u);
    e.y = static_cast<int>((((b >> 4) & 0xFu) * 0x00204081u) & 0x01010101u);
    e.z = static_cast<int>((((b >> 8) & 0xFu) * 0x00204081u) & 0x01010101u);
    e.w = static_cast<int>((((b >> 12) & 0xFu) * 0x00204081u) & 0x01010101u);
    return e;
}

__host__ __device__ inline uint64_t gram_gt_index(uint64_t i, uint64_t j, uint64_t A) { return i * A - i * (i - 1) / 2 + (j - i); }

// 128 x 128 alleles per workgroup (tile pair I <= J) of eight wavefronts, 64 x 32 per wavefront (two 32 x 32 tiles, five planes each:
// 160 accumulator registers; two wavefronts per SIMD, one spreading bits while the other is in the matrix pipe); a range of
// column words per blockIdx.y; S[gt] += the wavefront's part
__global__ __launch_bounds__(512) void gram_mfma_kernel(const uint32_t* __restrict__ bits, uint64_t kw, const uint8_t* __restrict__ dig, uint64_t k_cap,
                                                        uint32_t A, uint32_t n_tiles, uint64_t n_words, uint64_t words_per_split,
                                                        unsigned long long* __restrict__ S) {
    uint32_t tp = blockIdx.x, I = 0;
    while (tp >= n_tiles - I) { tp -= n_tiles - I; I++; }
    const uint32_t J = I + tp;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t wi = wave >> 2, wj = wave & 3u;
    const uint32_t row0 = I * 128 + wi * 64, col0 = J * 128 + wj * 32;
    if (I == J && row0 > col0 + 31) return;                                     // entirely below the diagonal
    const uint32_t r = lane & 31u, h = lane >> 5;
    const uint64_t w_begin = static_cast<uint64_t>(blockIdx.y) * words_per_split, w_end = min(n_words, w_begin + words_per_split);
    // this lane's alleles: a row (A operand) per 32 x 32 tile and the column (B operand)
    const uint32_t ia0 = row0 + r, ia1 = row0 + 32 + r, ib = col0 + r;
    const bool ok0 = ia0 < A, ok1 = ia1 < A, okb = ib < A;
    const uint32_t* pa0 = bits + static_cast<uint64_t>(ok0 ? ia0 : 0u) * kw;
    const uint32_t* pa1 = bits + static_cast<uint64_t>(ok1 ? ia1 : 0u) * kw;
    const uint32_t* pb = bits + static_cast<uint64_t>(okb ? ib : 0u) * kw;
    const uint8_t* pd = dig + h * 16;
    v16i acc[2][GR_PLANES];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (uint32_t p = 0; p < GR_PLANES; p++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[a][p][e] = 0;
    const uint32_t sh = 16 * h;
    // four k-steps (128 columns) per 16-byte load of each of the three bit rows, the next four in flight while these are used
    // (rows are 4-word aligned: the column capacity is a multiple of 128, and so are the splits)
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    auto load4 = [&](const uint32_t* p, bool ok, uint64_t w) -> u4 {
        return ok && w < w_end ? *reinterpret_cast<const u4*>(p + w) : u4{0u, 0u, 0u, 0u};
    };
    u4 na0 = load4(pa0, ok0, w_begin), na1 = load4(pa1, ok1, w_begin), nb = load4(pb, okb, w_begin);
    // the digits of this lane half's 16 columns of a k-step, one k-step ahead
    v4i dgn[GR_PLANES];
#pragma unroll
    for (uint32_t p = 0; p < GR_PLANES; p++) dgn[p] = *reinterpret_cast<const v4i*>(pd + static_cast<uint64_t>(p) * k_cap + w_begin * 32);
    for (uint64_t w = w_begin; w < w_end; w += 4) {
        const u4 ca0 = na0, ca1 = na1, cb = nb;
        na0 = load4(pa0, ok0, w + 4); na1 = load4(pa1, ok1, w + 4); nb = load4(pb, okb, w + 4);
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            // the rows' columns as bytes of 0 / 0xFF, cut to the plane's digits (A operand); the column's as bytes of 0 / 1 (B operand, the
            // same for every plane). With the digits on the B side instead — one tile of it per wavefront, half the ANDs — the loop
            // was a third slower (measured: 45 ms against 34 ms at 131 072 x 4 096).
            const v4i s0 = spread16((ca0[q] >> sh) & 0xFFFFu), s1 = spread16((ca1[q] >> sh) & 0xFFFFu);
            const v4i ea0 = (s0 << 8) - s0, ea1 = (s1 << 8) - s1;
            const v4i eb = spread16((cb[q] >> sh) & 0xFFFFu);
            v4i dg[GR_PLANES];
#pragma unroll
            for (uint32_t p = 0; p < GR_PLANES; p++) dg[p] = dgn[p];
            // (the digit planes are k_cap long, a multiple of 128 columns: one k-step behind the last word of a row is still inside)
            const uint64_t wn = min(w + q + 1, k_cap / 32 - 1);
#pragma unroll
            for (uint32_t p = 0; p < GR_PLANES; p++) dgn[p] = *reinterpret_cast<const v4i*>(pd + static_cast<uint64_t>(p) * k_cap + wn * 32);
#pragma unroll
            for (uint32_t p = 0; p < GR_PLANES; p++) {
                acc[0][p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ea0 & dg[p], eb, acc[0][p], 0, 0, 0);
                acc[1][p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ea1 & dg[p], eb, acc[1][p], 0, 0, 0);
            }
        }
    }
    // C/D layout of the 32 x 32 forms: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const uint32_t j = col0 + r;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const uint32_t i = row0 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (i > j || j >= A) continue;
            unsigned long long sum = 0;
#pragma unroll
            for (uint32_t p = 0; p < GR_PLANES; p++) sum += static_cast<unsigned long long>(static_cast<uint32_t>(acc[a][p][e])) << (7 * p);
            if (sum) atomicAdd(&S[gram_gt_index(i, j, A)], sum);
        }
}

__global__ __launch_bounds__(256) void gram_finish_kernel(const unsigned long long* __restrict__ S, uint64_t G, const double* __restrict__ c_part,
                                                          uint32_t n_parts, const int* __restrict__ f, const double* __restrict__ extra,
                                                          double* __restrict__ scores) {
    __shared__ double c_sh;
    if (threadIdx.x == 0) {
        double c = 0.0;
        for (uint32_t i = 0; i < n_parts; i++) c += c_part[i];                  // blocks in order: reproducible
        c_sh = c;
    }
    __syncthreads();
    const uint64_t g = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (g >= G) return;
    const double s = ldexp(static_cast<double>(S[g]), -*f);
    scores[g] = (c_sh - s) + (extra ? extra[g] : 0.0);
}

// the rows the Gram form leaves out, in order, as a matrix of their own for the f64 kernel
__global__ __launch_bounds__(256) void gram_residual_rows_kernel(const uint8_t* __restrict__ residual, uint64_t n_rows, uint32_t* __restrict__ list,
                                                                 unsigned long long* __restrict__ counters) {
    // one workgroup, rows in order
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint64_t r0 = 0; r0 < n_rows; r0 += 256) {
        const uint64_t r = r0 + threadIdx.x;
        const uint32_t f = r < n_rows && residual[r] ? 1u : 0u;
        const unsigned long long m = __ballot(f);
        const uint32_t before = static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)));
        if (lane == 0) wave_tot[wave] = static_cast<uint32_t>(__popcll(m));
        __syncthreads();
        uint32_t o = carry;
        for (uint32_t q = 0; q < wave; q++) o += wave_tot[q];
        if (f) list[o + before] = static_cast<uint32_t>(r);
        __syncthreads();
        if (threadIdx.x == 0) carry += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) counters[2] = carry;
}
__global__ __launch_bounds__(256) void gram_copy_rows_kernel(const double* __restrict__ M, uint32_t A, const uint32_t* __restrict__ list, uint32_t n,
                                                             double* __restrict__ out) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= static_cast<uint64_t>(n) * A) return;
    out[i] = M[static_cast<uint64_t>(list[i / A]) * A + i % A];
}

template <uint32_t VPL>
void launch_levels(hipStream_t s, uint32_t blocks, const double* M, uint64_t R, uint32_t A, uint64_t k_cap, uint64_t kw, uint32_t* bits, double* delta,
                   double* c_part, uint8_t* residual, unsigned long long* counters, uint32_t lmax) {
    hipLaunchKernelGGL(gram_levels_kernel<VPL>, dim3(blocks), dim3(256), 0, s, M, R, A, k_cap, kw, bits, delta, c_part, residual, counters, lmax);
}

}  // namespace

void launch_prefilter_tile(lcty_reads* reads, const double* M, uint64_t R, double* d_scores_out);      // lcty_prefilter.hip

// true: reads->d_scores holds the scores of all genotypes. false: this batch is not for the Gram form (few alleles, too many
// levels per row, ...) and nothing was written.
bool launch_prefilter_gram(lcty_reads* reads) {
    lcty_ctx* ctx = reads->ctx;
    const uint32_t A = reads->locus->n_alleles;
    const uint64_t R = reads->n_pairs, G = count_genotypes(A, 2);
    const int64_t want = ctx->knob("prefilter_gram", -1);
    if (want == 0 || A > 4096 || R == 0 || R >= (1ull << 32)) return false;
    if (want < 0 && A < 512) return false;                                       // the f64 tile kernel is a few milliseconds there
    hipStream_t s = ctx->stream;
    auto& B = reads->gram;
    // room for the columns: six per row on average (rows have 3.5 - 4.1), whole words per block of rows
    const uint64_t n_blocks = (R + GR_RB - 1) / GR_RB;
    const uint64_t k_cap = ((R * static_cast<uint64_t>(std::max<int64_t>(1, ctx->knob("prefilter_gram_cols", 6))) + n_blocks * 32 + 127) / 128) * 128;
    if (k_cap >= (1ull << 31) / 127) return false;                               // i32 accumulators: 127 * columns < 2^31
    const uint64_t kw = k_cap / 32;
    B.bits.ensure(static_cast<uint64_t>(A) * kw); B.delta.ensure(k_cap); B.dig.ensure(k_cap * GR_PLANES);
    B.c_part.ensure(n_blocks); B.residual.ensure(R); B.res_list.ensure(R); B.counters.ensure(4); B.f.ensure(1);
    B.counters.zero(s);
    LCTY_HIP(hipMemsetAsync(B.delta.p, 0, k_cap * sizeof(double), s));
    if (reads->d_scores.n < G) reads->d_scores.alloc(G);
    reads->n_scores = G;
    const double* M = reads->d_matrix.p;
    const uint32_t vpl = (A + 63) / 64;
    const uint32_t blocks = static_cast<uint32_t>(n_blocks);
    unsigned long long cnt[4] = {0, 0, 0, 0};
    const uint32_t lmax = static_cast<uint32_t>(std::min<int64_t>(GR_LMAX, std::max<int64_t>(1, ctx->knob("prefilter_gram_levels", GR_LMAX))));
    ctx->timed(LCTY_K_PREFILTER, [&] {
        if (vpl <= 4) launch_levels<4>(s, blocks, M, R, A, k_cap, kw, B.bits.p, B.delta.p, B.c_part.p, B.residual.p, B.counters.p, lmax);
        else if (vpl <= 8) launch_levels<8>(s, blocks, M, R, A, k_cap, kw, B.bits.p, B.delta.p, B.c_part.p, B.residual.p, B.counters.p, lmax);
        else if (vpl <= 16) launch_levels<16>(s, blocks, M, R, A, k_cap, kw, B.bits.p, B.delta.p, B.c_part.p, B.residual.p, B.counters.p, lmax);
        else if (vpl <= 32) launch_levels<32>(s, blocks, M, R, A, k_cap, kw, B.bits.p, B.delta.p, B.c_part.p, B.residual.p, B.counters.p, lmax);
        else launch_levels<64>(s, blocks, M, R, A, k_cap, kw, B.bits.p, B.delta.p, B.c_part.p, B.residual.p, B.counters.p, lmax);
        hipLaunchKernelGGL(gram_residual_rows_kernel, dim3(1), dim3(256), 0, s, B.residual.p, R, B.res_list.p, B.counters.p);
    }, s);
    LCTY_HIP(hipGetLastError());
    B.counters.download(cnt, 4, s);
    LCTY_HIP(hipStreamSynchronize(s));
    if (cnt[1]) return false;                                                    // more columns than room: the f64 kernel takes the batch
    const uint64_t n_cols = cnt[0], n_res = cnt[2];
    if (want < 0 && n_res * 4 > R) return false;                                 // mostly many-valued rows (long reads): not this form
    // the residual rows through the f64 kernel
    const double* extra = nullptr;
    if (n_res) {
        B.res_rows.ensure(n_res * A); B.res_scores.ensure(G);
        const uint64_t n = n_res * A;
        hipLaunchKernelGGL(gram_copy_rows_kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, s, M, A,
                           B.res_list.p, static_cast<uint32_t>(n_res), B.res_rows.p);
        launch_prefilter_tile(reads, B.res_rows.p, n_res, B.res_scores.p);
        extra = B.res_scores.p;
    }
    B.S.ensure(G);
    B.S.zero(s);
    const uint32_t n_tiles = (A + 127) / 128, n_tp = n_tiles * (n_tiles + 1) / 2;
    const uint64_t n_words = n_cols / 32;
    const uint32_t cus = static_cast<uint32_t>(ctx->props.multiProcessorCount);
    uint64_t splits = std::max<uint64_t>(1, std::min<uint64_t>((4ull * cus + n_tp - 1) / n_tp, std::max<uint64_t>(1, n_words / 64)));
    const uint64_t per = (std::max<uint64_t>(1, (n_words + splits - 1) / splits) + 3) / 4 * 4;
    splits = std::max<uint64_t>(1, (n_words + per - 1) / per);
    ctx->timed(LCTY_K_PREFILTER, [&] {
        // digits of whole groups of four words: the contraction reads 128 columns at a time, the weights behind the last column are 0
        hipLaunchKernelGGL(gram_digits_kernel, dim3(static_cast<uint32_t>(std::max<uint64_t>(1, (n_cols + 127) / 128 * 128 / 256 + 1))), dim3(256), 0, s, B.delta.p, std::min<uint64_t>((n_cols + 127) / 128 * 128, k_cap), k_cap,
                           B.counters.p, B.dig.p, B.f.p);
        if (n_words)
            hipLaunchKernelGGL(gram_mfma_kernel, dim3(n_tp, static_cast<uint32_t>(splits)), dim3(512), 0, s, B.bits.p, kw, B.dig.p, k_cap, A, n_tiles,
                               n_words, per, B.S.p);
        hipLaunchKernelGGL(gram_finish_kernel, dim3(static_cast<uint32_t>((G + 255) / 256)), dim3(256), 0, s, B.S.p, G, B.c_part.p, blocks, B.f.p, extra,
                           reads->d_scores.p);
    }, s);
    LCTY_HIP(hipGetLastError());
    return true;
}

}  // namespace lcty
