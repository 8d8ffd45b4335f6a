// lcty_solve.hip — the solver stages of `locityper genotype` on gfx950.
//
// Data the stages work on (built once per scored batch, all in HBM):
//   location table  T[contig][good read]  — allele-major transposition of the (pair, contig) index the scoring
//        kernel leaves behind: ln-probability and the two alignment middles of the best pair-alignment on that
//        contig, the number of pair-alignments there, and where the others sit in the pair-alignment arena.
//        A chain of genotype (a, b, ..) streams rows a, b, .. with coalesced loads; a single move needs one
//        16-byte gather per contig.
//   depth table     lut[GC bin][depth]    — DistrCache (src/model/distr_cache.rs:61-92) extended past the
//        256 entries of the reference's LinearCache (src/math/distr/lincache.rs:41-48) with the same
//        BayesCalc::ln_pmf (bayes.rs:27-35) evaluated on the device, so that one deep window costs one
//        L2 gather instead of (2 + n_alt) lgamma calls.
//
// Kernels:
//   build_loc_table_kernel   K11 input: GenotypeAlignments::new (assgn.rs:41-84) is NOT materialised; the possible
//        locations of a read on a genotype (extend_read_gt_alns, windows.rs:762-797) are re-derived on demand as a
//        merge of <= ploidy sorted runs plus "both mates unmapped".
//   solve_init_kernel<P>     one 256-thread workgroup per (genotype, attempt) chain, streaming the table rows of its alleles:
//        K12 apply_tweak (assgn.rs:127-151): window distributions (generate_windows, windows.rs:478-486;
//            neighb_info 439-445; get_distribution distr_cache.rs:83-92)
//        K13 ReadAssignment::try_new (assgn.rs:199-226): initial location of every read, depth histogram
//            (LDS atomics), alignment likelihood, and the chain's ordered list of non-trivial reads as 32-byte
//            RECORDS: everything a move of that read can need — its possible locations on the genotype in order
//            (ln-probability, tweaked windows) and the current one. The state-independent half of every later move
//            (two table gathers, the merge of the contigs' runs, the tweak draws) is done here once per (chain, read),
//            coalesced, instead of once per candidate behind three dependent random gathers.
//   greedy_loop_kernel<LPC>  K14 Greedy (stoch.rs:81-120): 64 / LPC chains per wavefront, a row of LPC lanes per chain,
//            lane = candidate read of the iteration; one 32-byte record gather per candidate, issued two iterations
//            ahead (the greedy random stream does not depend on the moves); window depths of the chain in LDS.
//   anneal_loop_kernel       K14 SimAnneal (stoch.rs:195-245): a chain wavefront plus a staging wavefront that runs the
//            random stream ahead and stages the records of the coming draws in an LDS ring.
//
// Randomness is the injected per-chain seed described in oracle/lcty_oracle.h (the reference's rand adaptors
// are not in its tree): counter-based draws for tweaks / random starts, xoshiro256++ for the solver loop.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <memory>
#include <numeric>
#include <thread>
#include <type_traits>

#include "lcty_exact.hpp"
#include "lcty_objects.hpp"

namespace lcty {

// Forms of the greedy loop (template parameter of the kernel; lcty_ctx_set_knob "solve_greedy_form" picks among the compiled ones):
//   32  (diagnostic) the phases of an iteration timed with the shader clock, summed per wavefront into SolveView::dbg
// Tried in round 4 and dropped (profiles/r04_greedy_forms_and_phases.txt): duplicate check / row maximum through LDS, the depth table
// as pairs, 16-byte record loads, a lane per alternative (further alternatives of a read on spare lanes of its row) — none moved the
// loop. What an iteration costs is the cache lines it has to bring into the CU — one record line from HBM and four table lines from
// the L2 per candidate, ~250 per wavefront-iteration, ~165 in flight per CU — whatever the instructions around them look like.

constexpr uint32_t MAXP = 4;                  // ploidy handled by the device solver
constexpr uint32_t NONE32S = 0xFFFFFFFFu;
constexpr uint32_t MID_NONE24 = 0xFFFFFFu;    // "mate unmapped" inside the location table
constexpr uint64_t WINDOW_KEY_XOR = 0xD1B54A32D192ED03ull;
constexpr uint64_t INIT_KEY_XOR = 0x8CB92BA72F3D8DD7ull;

// exact n / d and n % d for every 32-bit n by multiplication (Granlund & Montgomery); d is a locus constant
struct FastDiv {
    uint32_t d, m, s1, s2;
    __host__ static FastDiv make(uint32_t d) {
        FastDiv f{d, 0, 0, 0};
        uint32_t l = 0;
        while ((1ull << l) < d) l++;
        f.m = static_cast<uint32_t>(((1ull << 32) * ((1ull << l) - d)) / d + 1);
        f.s1 = l < 1 ? l : 1; f.s2 = l > 0 ? l - 1 : 0;
        return f;
    }
    __device__ __forceinline__ uint32_t div(uint32_t n) const {
        const uint32_t t = __umulhi(m, n);
        return (t + ((n - t) >> s1)) >> s2;
    }
    __device__ __forceinline__ uint32_t mod(uint32_t n) const { return n - div(n) * d; }
};

// one (contig, good read) cell of the location table: everything a move needs from this contig in one 32-byte gather
struct __attribute__((aligned(32))) LocEntry {
    double lp;                      // best pair-alignment of the read pair on the contig, -inf = none
    uint32_t m1n;                   // its first middle (24 bit, MID_NONE24 = unmapped mate) | number of pair-alignments << 24
    uint32_t m2;                    // its second middle
    double unm;                     // "both mates unmapped" probability of the read pair (same in every row)
    uint32_t ext;                   // arena index of the second pair-alignment on this contig
    uint32_t _pad;
};
static_assert(sizeof(LocEntry) == 32, "LocEntry layout");
// The table the chains stream is LEAN: 16 bytes per (contig, good read) — what every read needs — with the rest in side arrays:
// `unm` is a property of the read (one f64 per good read, not per cell), `ext` matters only to cells with more than one
// pair-alignment (a u32 per cell, gathered by the few lanes that need it). solve_init_kernel reads 2 x 16 + 8 B per (chain, read)
// instead of 2 x 32 B. LocEntry above is the form in which rows TRAVEL between shards (pack_rows_kernel / place_rows_kernel).
struct __attribute__((aligned(16))) LocCell {
    double lp;                      // best pair-alignment of the read pair on the contig, -inf = none
    uint32_t m1n;                   // its first middle (24 bit, MID_NONE24 = unmapped mate) | number of pair-alignments << 24
    uint32_t m2;                    // its second middle
};
static_assert(sizeof(LocCell) == 16, "LocCell layout");

// ChainRec (lcty_common.hpp): one non-trivial read of one chain, all a move can need in one 32-byte gather. Locations in the
// order of extend_read_gt_alns (windows.rs:793: ln-probability descending, ties in push order); windows after apply_tweak.

struct SolveView {
    // locus
    uint32_t A, window, left_padding, tweak;
    FastDiv by_window, by_tweak;    // window; 2 * tweak + 1
    double min_weight, prob_diff, depth_contrib, aln_contrib;
    const uint32_t* n_windows;
    const uint32_t* reg_start;
    const uint32_t* allele_len;
    const uint32_t* ci_off;
    const uint8_t* gc;
    const double* win_weight;
    // the two factors of a window weight as tables over the counts they are functions of (locus without explicit weights):
    // win_weight[i] == wk[uniq_cnt[i]] * wc[compl_cnt[i]] bit for bit; wk[n_wk - 1] = 0 stands for "trivial distribution"
    const uint32_t* uniq_cnt; const uint16_t* compl_cnt;
    const double* wk; const double* wc;
    uint32_t n_wk, n_wc;            // 0: no tables (the greedy loop then gathers the weights)
    const double* lut;              // [LCTY_GC_BINS][lut_depth]
    uint32_t lut_depth, lut_shift;  // lut_depth = 1 << lut_shift
    const DepthNB* depth_nb;
    uint32_t n_alt;
    // reads
    uint32_t n_good;
    uint64_t ngp;                   // row stride of the location table (n_good rounded up to 64)
    const LocCell* table;           // [A][ngp], or [rows][ngp] with row_of when only the rows of some alleles are held
    const uint32_t* table_ext;      // same shape: arena index of the second pair-alignment of the cell (cells with more than one)
    const double* table_unm;        // [ngp] "both mates unmapped" probability of every good read pair
    const uint16_t* row_of;         // allele -> row of `table` (NULL: the allele itself)
    const PairAlnDev* pa;
    // chains
    const uint16_t* genotypes;      // [n_gt][ploidy]
    uint32_t ploidy, attempts;
    const uint64_t* seeds;          // [n_chains]
    const double* priors;           // [n_gt] or null
    lcty_solver solver;
    ChainRec* recs;                 // [n_chains][rstride] the chain's non-trivial reads in read order, in INIT_SEGS segments (RecList)
    uint32_t seg_reads;             // reads (and record places) per segment; rstride = INIT_SEGS * seg_reads >= n_good
    uint64_t rstride;
    uint32_t* c_seg;                // [n_chains][4] non-trivial reads in front of segment 0..3 (c_seg[.][0] = 0)
    ExtraLoc* extra;                // [n_chains][extra_cap] locations 2.. of reads with more than two
    uint32_t extra_cap;
    uint32_t* c_totw;               // [n_chains] windows of the chain's genotype (2 + sum of n_windows)
    uint32_t wstride;               // per-chain stride of the window arrays (>= 2 + ploidy * max n_windows)
    double* c_ww;                   // [n_chains][wstride] window weights (0 = trivial distribution)
    uint32_t* c_uc;                 // [n_chains][wstride] index into wk | index into wc << 16 of the window (with the tables)
    uint8_t* c_gc;                  // [n_chains][wstride]
    uint32_t* c_depth;              // [n_chains][wstride]
    uint32_t* c_nnt;                // [n_chains]
    double* c_aln;                  // [n_chains] alignment likelihood after K13
    double* liks;                   // [n_chains] prior + likelihood
    double* parts;                  // [n_chains][4] aln_lik, depth_lik, solver iterations, accepted moves (diagnostics)
    uint32_t* overflow;             // set when a window got deeper than the depth table (the host widens it and repeats)
    double* dbg;                    // [wavefronts][8] clock cycles per phase of the greedy iteration (form 32 only)
    uint32_t prio_mode;             // issue priority of the loop kernels: 0 annealing wavefronts first (default), 1 greedy first, 2 none (knob solve_prio_mode)
};

// a wave-uniform 64-bit value, told to the compiler as such (it then lives in scalar registers)
__device__ __forceinline__ uint64_t uniform64(uint64_t x) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(x));
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(x >> 32));
    return (static_cast<uint64_t>(hi) << 32) | lo;
}
// ---- randomness (definitions shared with oracle/lcty_oracle_solve.c) ----
__device__ __forceinline__ uint64_t counter_u64(uint64_t key, uint64_t i) {
    uint64_t z = key + (i + 1) * 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
struct Xoshiro {
    uint64_t s0, s1, s2, s3;
    __device__ __forceinline__ void seed(uint64_t x) {         // seed_from_u64: SplitMix64 fill (ext/rand.rs:3-22)
        s0 = counter_u64(x, 0); s1 = counter_u64(x, 1); s2 = counter_u64(x, 2); s3 = counter_u64(x, 3);
    }
    __device__ __forceinline__ uint64_t next() {               // xoshiro256++
        const uint64_t sum = s0 + s3;
        const uint64_t result = ((sum << 23) | (sum >> 41)) + s0;
        const uint64_t t = s1 << 17;
        s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t; s3 = (s3 << 45) | (s3 >> 19);
        return result;
    }
    __device__ __forceinline__ uint64_t below(uint64_t n) { return __umul64hi(next(), n); }
    __device__ __forceinline__ double f64() { return static_cast<double>(next() >> 11) * (1.0 / 9007199254740992.0); }
};

// The ordered list of a chain's non-trivial reads (assgn.rs:61-63), as solve_init_kernel leaves it: the reads of the locus are cut
// into INIT_SEGS contiguous ranges, one per wavefront of the initialisation, and every wavefront compacts ITS range into ITS part
// of the chain's record array (part k starts at k * seg_reads) — no exchange between the wavefronts, no barrier per block of
// reads. The list in read order is the parts one after the other: entry s is record s - cum[k] of part k, where k is the part
// with cum[k] <= s < cum[k + 1]. cum: four words (cum[0] = 0) in LDS or global memory.
constexpr uint32_t INIT_SEGS = 4;
struct RecList {
    ChainRec* base; const uint32_t* cum; uint32_t seg_reads;
    uint32_t r1 = 0, r2 = 0, r3 = 0; bool in_regs = false;                  // the three bounds held by the lane itself (greedy loop)
    __device__ __forceinline__ uint32_t place(uint32_t s) const {
        const uint32_t c1 = in_regs ? r1 : cum[1], c2 = in_regs ? r2 : cum[2], c3 = in_regs ? r3 : cum[3];
        const uint32_t k = (s >= c1 ? 1u : 0u) + (s >= c2 ? 1u : 0u) + (s >= c3 ? 1u : 0u);
        const uint32_t before = s >= c3 ? c3 : s >= c2 ? c2 : s >= c1 ? c1 : 0u;
        return s - before + k * seg_reads;
    }
    __device__ __forceinline__ ChainRec& operator[](uint32_t s) const { return base[place(s)]; }
};

// ---- K11 input: allele-major location table ----
__global__ __launch_bounds__(256) void build_loc_table_kernel(const uint32_t* __restrict__ good_ix, uint32_t n_good, uint64_t ngp,
                                                              uint32_t A, const uint32_t* __restrict__ pa_idx,
                                                              const uint64_t* __restrict__ pa_off, const PairAlnDev* __restrict__ pa,
                                                              const double* __restrict__ unmapped, LocCell* __restrict__ table,
                                                              uint32_t* __restrict__ table_ext, double* __restrict__ table_unm,
                                                              uint32_t* __restrict__ err) {
    // 32 reads x 32 contigs per workgroup: the (pair, contig) index is read along contigs, the table written along reads
    __shared__ LocCell tile[32][33];
    __shared__ uint32_t tile_ext[32][33];
    const uint32_t g0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const uint32_t tx = threadIdx.x & 31u, ty = threadIdx.x >> 5;
    if (blockIdx.y == 0 && threadIdx.x < 32) {
        const uint32_t g = g0 + threadIdx.x;
        if (g < ngp) table_unm[g] = g < n_good ? unmapped[good_ix[g]] : 0.0;
    }
    for (uint32_t i = ty; i < 32; i += 8) {
        const uint32_t g = g0 + i, c = c0 + tx;
        LocCell e{-INFINITY, MID_NONE24, NONE32S};
        uint32_t ext = 0;
        if (g < n_good && c < A) {
            const uint32_t r = good_ix[g];
            const uint32_t idx = pa_idx[static_cast<uint64_t>(r) * A + c];
            const uint32_t cnt = idx >> 24;
            if (cnt) {
                const uint64_t at = pa_off[r] + (idx & 0xFFFFFFu);
                const PairAlnDev p = pa[at];
                if ((p.mid1 != NONE32S && p.mid1 >= MID_NONE24) || at + cnt > 0xFFFFFFFFull)
                    atomicMax(err, static_cast<uint32_t>(LCTY_ERR_UNSUPPORTED));
                e.lp = p.ln_prob;
                e.m1n = (p.mid1 == NONE32S ? MID_NONE24 : p.mid1) | (cnt << 24);
                e.m2 = p.mid2;
                ext = static_cast<uint32_t>(at + 1);
            }
        }
        tile[i][tx] = e; tile_ext[i][tx] = ext;
    }
    __syncthreads();
    for (uint32_t j = ty; j < 32; j += 8) {
        const uint32_t c = c0 + j, g = g0 + tx;
        if (c < A && g < ngp) { table[static_cast<uint64_t>(c) * ngp + g] = tile[tx][j]; table_ext[static_cast<uint64_t>(c) * ngp + g] = tile_ext[tx][j]; }
    }
}

// ---- the location-table rows of some alleles over the reads of SEVERAL batches (shards of one locus' reads, SURVEY 8e level 2):
// per shard the rows are packed — cells of the wanted alleles, the pair-alignments behind the first one of a cell compacted into
// a run of the shard's own — and then laid side by side, shard after shard, into one table whose cells point into one array of
// further pair-alignments. `ext` of a packed cell: index into the shard's run.
__global__ __launch_bounds__(256) void pack_rows_count_kernel(const LocCell* __restrict__ table, uint64_t ngp, uint32_t n_good,
                                                              const uint16_t* __restrict__ alleles, uint32_t n_rows,
                                                              unsigned long long* __restrict__ total) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
    uint32_t mine = 0;
    if (i < static_cast<uint64_t>(n_rows) * n_good) {
        const uint32_t u = static_cast<uint32_t>(i / n_good), g = static_cast<uint32_t>(i % n_good);
        const uint32_t cnt = table[static_cast<uint64_t>(alleles[u]) * ngp + g].m1n >> 24;
        mine = cnt > 1 ? cnt - 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) mine += static_cast<uint32_t>(__shfl_xor(static_cast<int>(mine), o));
    if ((threadIdx.x & 63u) == 0 && mine) atomicAdd(total, static_cast<unsigned long long>(mine));
}
__global__ __launch_bounds__(256) void pack_rows_kernel(const LocCell* __restrict__ table, const uint32_t* __restrict__ table_ext,
                                                        const double* __restrict__ table_unm, uint64_t ngp, uint32_t n_good,
                                                        const uint16_t* __restrict__ alleles, uint32_t n_rows, const PairAlnDev* __restrict__ pa,
                                                        LocEntry* __restrict__ cells, uint64_t out_stride, PairAlnDev* __restrict__ extras,
                                                        unsigned long long* __restrict__ cursor) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= static_cast<uint64_t>(n_rows) * out_stride) return;
    const uint32_t u = static_cast<uint32_t>(i / out_stride), g = static_cast<uint32_t>(i % out_stride);
    LocEntry e{-INFINITY, MID_NONE24, NONE32S, 0.0, 0u, 0u};
    if (g < n_good) {
        const uint64_t at_cell = static_cast<uint64_t>(alleles[u]) * ngp + g;
        const LocCell c = table[at_cell];
        e.lp = c.lp; e.m1n = c.m1n; e.m2 = c.m2; e.unm = table_unm[g];
        const uint32_t cnt = e.m1n >> 24;
        if (cnt > 1) {
            const uint32_t src = table_ext[at_cell];
            const unsigned long long at = atomicAdd(cursor, static_cast<unsigned long long>(cnt - 1));
            for (uint32_t k = 0; k + 1 < cnt; k++) extras[at + k] = pa[src + k];
            e.ext = static_cast<uint32_t>(at);
        } else e.ext = 0;
    }
    cells[i] = e;
}
// one shard's packed rows into the gathered table: full[u][first + g] = cells[u][g], `ext` moved by where the shard's run starts
__global__ __launch_bounds__(256) void place_rows_kernel(const LocEntry* __restrict__ cells, uint64_t in_stride, uint32_t n_good, uint32_t n_rows,
                                                         uint32_t ext_base, LocCell* __restrict__ full, uint32_t* __restrict__ full_ext,
                                                         double* __restrict__ full_unm, bool write_unm, uint64_t full_stride, uint64_t first) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= static_cast<uint64_t>(n_rows) * n_good) return;
    const uint32_t u = static_cast<uint32_t>(i / n_good), g = static_cast<uint32_t>(i % n_good);
    const LocEntry e = cells[static_cast<uint64_t>(u) * in_stride + g];
    const uint64_t at = static_cast<uint64_t>(u) * full_stride + first + g;
    full[at] = LocCell{e.lp, e.m1n, e.m2};
    full_ext[at] = (e.m1n >> 24) > 1 ? e.ext + ext_base : 0u;
    if (write_unm && u == 0) full_unm[first + g] = e.unm;
}
__global__ __launch_bounds__(256) void pad_rows_kernel(LocCell* __restrict__ full, uint32_t* __restrict__ full_ext, double* __restrict__ full_unm,
                                                       uint64_t full_stride, uint64_t from, uint32_t n_rows) {
    const uint64_t width = full_stride - from;
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= static_cast<uint64_t>(n_rows) * width) return;
    const uint64_t at = (i / width) * full_stride + from + i % width;
    full[at] = LocCell{-INFINITY, MID_NONE24, NONE32S};
    full_ext[at] = 0u;
    if (i / width == 0) full_unm[from + i % width] = 0.0;
}

// BayesCalc::ln_pmf evaluated directly: bayes.rs:27-35 with Ln::map_sum_init (math/mod.rs:80-94)
__device__ __noinline__ double bayes_ln_pmf_direct(const DepthNB* nb, uint32_t n_alt, uint32_t depth) {
    const double x = static_cast<double>(depth);
    const double lg1 = lgamma(x + 1.0);
    const double null_prob = nb->lnpmf_const[0] + lgamma(nb->n[0] + x) - lg1 + x * nb->lnq;
    if (n_alt == 0) return null_prob - null_prob;
    double v[LCTY_MAX_ALT_CN];
    double m = null_prob;
    for (uint32_t i = 0; i < n_alt; i++) {
        v[i] = nb->lnpmf_const[i + 1] + lgamma(nb->n[i + 1] + x) - lg1 + x * nb->lnq;
        m = fmax(m, v[i]);
    }
    double sum_prob;
    if (n_alt == 1) {                                                       // Ln::add (math/mod.rs:29-35)
        const double a = null_prob, b = v[0];
        if (a >= b) sum_prob = b == -INFINITY ? a : b + log1p(exp(a - b));
        else sum_prob = a == -INFINITY ? b : a + log1p(exp(b - a));
    } else if (isinf(m)) {
        sum_prob = m;
    } else {
        double s = exp(null_prob - m);
        for (uint32_t i = 0; i < n_alt; i++) s += exp(v[i] - m);
        sum_prob = m + log(s);
    }
    return null_prob - sum_prob;
}

// depth table: the first LCTY_DEPTH_CACHE columns are the locus' LinearCache, the others the same formula on the device
__global__ __launch_bounds__(256) void build_depth_table_kernel(const double* __restrict__ cache, const DepthNB* __restrict__ nb,
                                                                uint32_t n_alt, uint32_t lut_depth, double* __restrict__ lut) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= LCTY_GC_BINS * lut_depth) return;
    const uint32_t g = i / lut_depth, d = i % lut_depth;
    lut[i] = d < LCTY_DEPTH_CACHE ? cache[g * LCTY_DEPTH_CACHE + d] : bayes_ln_pmf_direct(nb + g, n_alt, d);
}

// ---- the genotype of a chain: GenotypeWindows (windows.rs:709-739) ----
template <uint32_t P>
struct Geno {
    uint32_t id[P], row[P], shift[P], reg_start[P], reg_end[P];
    uint32_t total_w;
    __device__ __forceinline__ void init(const SolveView& V, uint32_t gi) {
        total_w = 2;                                                        // REG_WINDOW_SHIFT
#pragma unroll
        for (uint32_t p = 0; p < P; p++) {
            id[p] = V.genotypes[static_cast<uint64_t>(gi) * P + p];
            row[p] = V.row_of ? V.row_of[id[p]] : id[p];
            shift[p] = total_w;
            reg_start[p] = V.reg_start[id[p]];
            const uint32_t nw = V.n_windows[id[p]];
            reg_end[p] = reg_start[p] + nw * V.window;
            total_w += nw;
        }
    }
};

// ---- possible locations of one read on the genotype (extend_read_gt_alns, windows.rs:762-797) ----
template <uint32_t P>
struct Locs {
    double lp[P];                  // best pair-alignment on contig p
    uint32_t m1[P], m2[P];
    uint32_t ext[P];               // arena index of the next one
    uint32_t n[P];                 // pair-alignments on contig p that pass the threshold
    double unm;
    bool has_unm;
    uint32_t nw;
};

// from the cells of the genotype's rows (registers), the read's "unmapped" probability and where the cells sit in the table (the
// `ext` of a cell is fetched only by a lane whose cell holds more than one pair-alignment)
template <uint32_t P>
__device__ __forceinline__ void locs_from_cells(Locs<P>& L, const SolveView& V, const LocCell* cells, double unm, const uint64_t* cell_at) {
    uint32_t raw[P];
    double top = -INFINITY;
    L.unm = unm;
#pragma unroll
    for (uint32_t p = 0; p < P; p++) {
        const LocCell e = cells[p];
        L.lp[p] = e.lp;
        raw[p] = e.m1n >> 24;
        L.m1[p] = (e.m1n & MID_NONE24) == MID_NONE24 ? NONE32S : (e.m1n & MID_NONE24);
        L.m2[p] = e.m2;
        L.ext[p] = 0;
        top = fmax(top, e.lp);                                              // -inf where the contig has nothing
    }
    top = fmax(top, L.unm);
    const double thresh = top - V.prob_diff;                                // max(unm - d, best_i - d, ...) == max(...) - d
    L.nw = 0;
#pragma unroll
    for (uint32_t p = 0; p < P; p++) {
        uint32_t k = (raw[p] && L.lp[p] >= thresh) ? 1u : 0u;
        if (k && raw[p] > 1) {
            L.ext[p] = V.table_ext[cell_at[p]];
            while (k < raw[p] && V.pa[L.ext[p] + k - 1].ln_prob >= thresh) k++;
        }
        L.n[p] = k;
        L.nw += k;
    }
    L.has_unm = L.unm >= thresh;
    L.nw += L.has_unm;
}

template <uint32_t P>
__device__ __forceinline__ void locs_init(Locs<P>& L, const SolveView& V, uint32_t g, const Geno<P>& G) {
    LocCell cells[P];
    uint64_t at[P];
#pragma unroll
    for (uint32_t p = 0; p < P; p++) { at[p] = static_cast<uint64_t>(G.row[p]) * V.ngp + g; cells[p] = V.table[at[p]]; }
    locs_from_cells<P>(L, V, cells, V.table_unm[g], at);
}

struct LocOut {
    double lp;
    uint32_t mid1, mid2, cix;              // cix 0xFF: both mates unmapped
};

// locations in decreasing ln_prob, ties in push order (contig_ix ascending, then "unmapped") — windows.rs:793
template <uint32_t P>
struct LocIter {
    double lp[P];
    uint32_t m1[P], m2[P], cur[P];
    bool unm_left;
    __device__ __forceinline__ void start(const Locs<P>& L) {
#pragma unroll
        for (uint32_t p = 0; p < P; p++) { cur[p] = 0; lp[p] = L.lp[p]; m1[p] = L.m1[p]; m2[p] = L.m2[p]; }
        unm_left = L.has_unm;
    }
    __device__ __forceinline__ bool next(const Locs<P>& L, const SolveView& V, LocOut& o) {
        double best = -INFINITY;
        uint32_t bp = NONE32S;
#pragma unroll
        for (uint32_t p = 0; p < P; p++) {
            if (cur[p] < L.n[p] && (bp == NONE32S || lp[p] > best)) { best = lp[p]; bp = p; }
        }
        if (unm_left && (bp == NONE32S || L.unm > best)) {
            unm_left = false;
            o.lp = L.unm; o.mid1 = o.mid2 = NONE32S; o.cix = 0xFFu;
            return true;
        }
        if (bp == NONE32S) return false;
        o.cix = bp;
#pragma unroll
        for (uint32_t p = 0; p < P; p++) {
            if (p == bp) {
                o.lp = lp[p]; o.mid1 = m1[p]; o.mid2 = m2[p];
                cur[p]++;
                if (cur[p] < L.n[p]) {
                    const PairAlnDev e = V.pa[L.ext[p] + cur[p] - 1];
                    lp[p] = e.ln_prob; m1[p] = e.mid1; m2[p] = e.mid2;
                }
            }
        }
        return true;
    }
};

// get_shifted_window_ix + middle_window (windows.rs:62-68, 465-470) with define_windows_random (123-136)
template <uint32_t P>
__device__ __forceinline__ void loc_windows(const SolveView& V, const Geno<P>& G, const LocOut& o, uint64_t seed, uint32_t rp,
                                            uint32_t t, uint32_t* w1, uint32_t* w2) {
    if (o.cix == 0xFFu) { *w1 = 0; *w2 = 0; return; }
    uint32_t t1 = 0, t2 = 0;
    if (V.tweak) {
        const uint64_t r = counter_u64(seed, (static_cast<uint64_t>(rp) << 16) | t);
        t1 = V.by_tweak.mod(static_cast<uint32_t>(r >> 32));
        t2 = V.by_tweak.mod(static_cast<uint32_t>(r));
    }
    uint32_t sh = 0, rs = 0, re = 0;
#pragma unroll
    for (uint32_t p = 0; p < P; p++) if (p == o.cix) { sh = G.shift[p]; rs = G.reg_start[p]; re = G.reg_end[p]; }
    auto ix = [&](uint32_t mid, uint32_t tw) -> uint32_t {
        if (mid == NONE32S) return 0u;                                       // UNMAPPED_WINDOW
        const uint32_t m = mid + tw;
        return (rs <= m && m < re) ? V.by_window.div(m - rs) + sh : 1u;      // BOUNDARY_WINDOW
    };
    *w1 = ix(o.mid1, t1);
    *w2 = ix(o.mid2, t2);
}

struct Move {                  // ReassignmentTarget + what reassign() needs
    uint32_t rp, new_assgn, slot;
    uint32_t w1, w2, w3, w4;
    double lp_old, lp_new;
    double ddiff;              // depth_lik_diff(w1, w2, w3, w4) at the time the move was evaluated
};

// ---------------- K12 + K13: one 256-thread workgroup per chain ----------------
// a barrier that orders LDS traffic only: __syncthreads() also waits for every global load of the wavefront (one counter for loads and
// stores on this architecture), which would end the prefetch below at the first barrier
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// inclusive prefix sums of two values over the 256 threads of the workgroup (wave shuffles + one LDS exchange); *total = sum of all
__device__ __forceinline__ void block_prefix_excl2(uint32_t va, uint32_t vb, uint32_t lane, uint32_t wave, uint2* wave_sums, uint32_t* ea, uint32_t* eb,
                                                   uint32_t* total_a, uint32_t* total_b) {
    uint32_t ia = va, ib = vb;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t ua = static_cast<uint32_t>(__shfl_up(static_cast<int>(ia), o)), ub = static_cast<uint32_t>(__shfl_up(static_cast<int>(ib), o));
        if (lane >= static_cast<uint32_t>(o)) { ia += ua; ib += ub; }
    }
    if (lane == 63) wave_sums[wave] = make_uint2(ia, ib);
    lds_barrier();
    uint32_t ba = 0, bb = 0, ta = 0, tb = 0;
#pragma unroll
    for (uint32_t q = 0; q < 4; q++) { const uint2 w = wave_sums[q]; if (q < wave) { ba += w.x; bb += w.y; } ta += w.x; tb += w.y; }
    *total_a = ta; *total_b = tb;
    lds_barrier();
    *ea = ba + ia - va; *eb = bb + ib - vb;
}

template <uint32_t P>
__global__ __launch_bounds__(256) void solve_init_kernel(const SolveView V) {
    extern __shared__ __align__(16) uint8_t smem[];
    uint32_t* depth = reinterpret_cast<uint32_t*>(smem);                          // [wstride]
    double* red = reinterpret_cast<double*>(smem + ((static_cast<size_t>(V.wstride) * 4 + 15) & ~static_cast<size_t>(15)));   // [256]
    uint32_t* seg_cnt = reinterpret_cast<uint32_t*>(red + 256);                   // [4] records of every segment, [4] = further locations handed out
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t chain = blockIdx.x;
    const uint32_t gi = chain / V.attempts;
    const uint64_t seed = V.seeds[chain];
    Geno<P> G; G.init(V, gi);
    ChainRec* recs = V.recs + static_cast<uint64_t>(chain) * V.rstride + static_cast<uint64_t>(wave) * V.seg_reads;     // this wavefront's part
    ExtraLoc* extra = V.extra + static_cast<uint64_t>(chain) * V.extra_cap;
    double* ww = V.c_ww + static_cast<uint64_t>(chain) * V.wstride;
    uint8_t* wgc = V.c_gc + static_cast<uint64_t>(chain) * V.wstride;
    uint32_t* wuc = V.n_wk ? V.c_uc + static_cast<uint64_t>(chain) * V.wstride : nullptr;

    // K12: window distributions (apply_tweak, assgn.rs:140-150)
    for (uint32_t w = tid; w < G.total_w; w += 256) {
        depth[w] = 0;
        if (w == 0) seg_cnt[4] = 0;
        double weight = 0.0; uint32_t g = 0, uc = V.n_wk - 1;                      // windows 0 and 1 (unmapped / out of region): trivial
        if (w >= 2) {
            uint32_t allele = G.id[0], sh = G.shift[0], rs = G.reg_start[0];
#pragma unroll
            for (uint32_t q = 1; q < P; q++) if (w >= G.shift[q]) { allele = G.id[q]; sh = G.shift[q]; rs = G.reg_start[q]; }
            const uint32_t start = rs + (w - sh) * V.window, end = start + V.window;
            const uint32_t left = min(V.tweak, start), right = min(V.tweak, V.allele_len[allele] - end);
            const uint64_t r = counter_u64(seed ^ WINDOW_KEY_XOR, w);                 // rng.random_range(-left..=right)
            const int64_t off = -static_cast<int64_t>(left) + static_cast<int64_t>(__umul64hi(r, static_cast<uint64_t>(left + right + 1)));
            const uint32_t wstart = static_cast<uint32_t>(static_cast<int64_t>(start) + off);
            const uint32_t i = V.ci_off[allele] + (wstart > V.left_padding ? wstart - V.left_padding : 0u);
            weight = V.win_weight[i];
            g = V.gc[i];
            if (wuc) uc = V.uniq_cnt[i] | (static_cast<uint32_t>(V.compl_cnt[i]) << 16);
            if (weight < V.min_weight || weight < 1e-7) { weight = 0.0; g = 0; uc = V.n_wk - 1; }      // assgn.rs:144-148, distr_cache.rs:84
        }
        ww[w] = weight; wgc[w] = static_cast<uint8_t>(g);
        if (wuc) wuc[w] = uc;
    }
    __syncthreads();

    // K13: initial assignment, depth histogram, the records of the non-trivial reads. The wavefronts of the workgroup do not talk to
    // each other here: wavefront k takes the k-th contiguous range of the locus' reads (seg_reads of them) and compacts its non-trivial
    // ones, in read order, into the k-th part of the chain's record array (RecList): a ballot and a population count per 64 reads
    // instead of a workgroup prefix sum behind two barriers per 256 (those exchanges were 23 of the kernel's 116 ms).
    // Further locations (beyond a read's second) come out of one run per chain through an LDS counter: where in the run they sit
    // is nobody's business but the record's.
    const bool random_start = V.solver.kind == LCTY_SOLVER_ANNEAL || (V.solver.kind == LCTY_SOLVER_GREEDY && !V.solver.best_start);
    double aln_part = 0.0;
    uint32_t n_recs = 0;
    uint64_t row_at[P];
#pragma unroll
    for (uint32_t p = 0; p < P; p++) row_at[p] = static_cast<uint64_t>(G.row[p]) * V.ngp;
    const uint32_t seg_lo = min(wave * V.seg_reads, V.n_good), seg_hi = min(seg_lo + V.seg_reads, V.n_good);
    // the cells of the next 64 reads are requested before this block's are used
    LocCell nxt[P]; double nxt_unm = 0.0;
    if (seg_lo + lane < seg_hi) {
#pragma unroll
        for (uint32_t p = 0; p < P; p++) nxt[p] = V.table[row_at[p] + seg_lo + lane];
        nxt_unm = V.table_unm[seg_lo + lane];
    }
    for (uint32_t base = seg_lo; base < seg_hi; base += 64) {
        const uint32_t rp = base + lane;
        LocCell cur[P]; uint64_t at[P];
#pragma unroll
        for (uint32_t p = 0; p < P; p++) { cur[p] = nxt[p]; at[p] = row_at[p] + rp; }
        const double cur_unm = nxt_unm;
        if (rp + 64 < seg_hi) {
#pragma unroll
            for (uint32_t p = 0; p < P; p++) nxt[p] = V.table[row_at[p] + rp + 64];
            nxt_unm = V.table_unm[rp + 64];
        }
        Locs<P> L; L.nw = 0;
        uint32_t a0 = 0;
        if (rp < seg_hi) {
            locs_from_cells<P>(L, V, cur, cur_unm, at);
            if (L.nw > 255) atomicMax(V.overflow, 2u);                           // a record keeps the location in 8 bits
            if (L.nw > 1 && random_start)
                a0 = static_cast<uint32_t>(__umul64hi(counter_u64(seed ^ INIT_KEY_XOR, rp), static_cast<uint64_t>(L.nw)));
        }
        const bool nontrivial = L.nw > 1;
        const uint32_t n_extra = L.nw > 2 ? min(L.nw, 255u) - 2u : 0u;
        // ordered compaction of the non-trivial reads of this range (assgn.rs:61-63)
        const unsigned long long nt_mask = __ballot(nontrivial);
        const uint32_t slot = n_recs + static_cast<uint32_t>(__popcll(nt_mask & ((1ull << lane) - 1ull)));
        n_recs += static_cast<uint32_t>(__popcll(nt_mask));
        if (L.nw > 0) {
            uint32_t eix = 0;
            if (n_extra) eix = atomicAdd(&seg_cnt[4], n_extra);
            ChainRec rec; rec.rp_cur = rp | (a0 << 24); rec.meta = min(L.nw, 255u) | (eix << 8);
            rec.lp0 = rec.lp1 = 0.0; rec.win0 = rec.win1 = 0;
            const bool room = static_cast<uint64_t>(eix) + n_extra <= V.extra_cap;
            LocIter<P> it; it.start(L);
            LocOut o;
            for (uint32_t t = 0; t < min(L.nw, 255u) && it.next(L, V, o); t++) {
                uint32_t wa, wb;
                loc_windows(V, G, o, seed, rp, t, &wa, &wb);
                const uint32_t win = wa | (wb << 16);
                if (t == 0) { rec.lp0 = o.lp; rec.win0 = win; }
                else if (t == 1) { rec.lp1 = o.lp; rec.win1 = win; }
                else if (room) { ExtraLoc e; e.lp = o.lp; e.win = win; e._pad = 0; extra[eix + t - 2] = e; }
                if (t == a0) {
                    atomicAdd(&depth[wa], 1u);
                    atomicAdd(&depth[wb], 1u);
                    aln_part += o.lp;
                }
            }
            if (nontrivial) recs[slot] = rec;
        }
    }
    if (lane == 0) seg_cnt[wave] = n_recs;
    red[tid] = aln_part;
    __syncthreads();
    const uint32_t ex_total = seg_cnt[4];
    if (tid == 0) {
        if (ex_total > V.extra_cap) { atomicMax(V.overflow, 4u); atomicMax(V.overflow + 1, ex_total); }
        if (ex_total >= (1u << 24)) atomicMax(V.overflow, 2u);
    }
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    uint32_t* gdepth = V.c_depth + static_cast<uint64_t>(chain) * V.wstride;
    for (uint32_t w = tid; w < G.total_w; w += 256) gdepth[w] = depth[w];
    if (tid == 0) {
        uint32_t* cum = V.c_seg + static_cast<uint64_t>(chain) * 4;
        uint32_t run = 0;
        for (uint32_t k = 0; k < INIT_SEGS; k++) { cum[k] = run; run += seg_cnt[k]; }
        V.c_aln[chain] = red[0]; V.c_nnt[chain] = run; V.c_totw[chain] = G.total_w;
    }
}

// two windows of one location (see Chain::request_pair)
struct PairGather { int32_t c[2]; uint32_t dmax[2]; double weight[2], vnew[2], vold[2]; };
// the four terms of depth_lik_diff in its order of summation, ((t1 + t2) + t3) + t4, from the two halves; *deepest: the deepest live window
__device__ __forceinline__ double pair_term(const PairGather& g, int i, uint32_t* deepest) {
    const bool live = g.c[i] != 0 && g.weight[i] != 0.0;                     // c == 0: no change; weight 0: WindowDistr::TRIVIAL
    *deepest = max(*deepest, live ? g.dmax[i] : 0u);
    return live ? g.weight[i] * g.vnew[i] - g.weight[i] * g.vold[i] : 0.0;
}

// window state of one chain: depth (25 bit) | GC bin << 25 in LDS, weights in the chain's row of c_ww (L2)
constexpr uint32_t DEPTH_MASK = 0x1FFFFFFu;
struct Chain {
    const SolveView* V;
    uint32_t* wd;               // LDS: the chain's windows
    const double* ww;           // window weights: LDS (annealing) or the chain's row of c_ww
    // WindowDistr::ln_prob (distr_cache.rs:34-39) through the depth table
    __device__ __forceinline__ double wlp(uint32_t w, uint32_t g, uint32_t d) const {
        const double weight = ww[w];
        if (weight == 0.0) return 0.0;                                      // WindowDistr::TRIVIAL
        if (d >= V->lut_depth) { atomicMax(V->overflow, 1u); return 0.0; }  // every chain of the batch is repeated
        return weight * V->lut[g * V->lut_depth + d];
    }
    __device__ __forceinline__ double wlp_at(uint32_t w) const { return wlp(w, wd[w] >> 25, wd[w] & DEPTH_MASK); }
    // depth_lik_diff (assgn.rs:259-284) = sum of atomic_depth_lik_diff (244-254) over the four windows, in two halves: `request`
    // reads the windows' depths (LDS) and ISSUES the twelve gathers (table entry at the old and the new depth, window weight);
    // `finish` uses them. Between the two a caller can issue further loads: the wait for these twelve then leaves those in flight.
    struct DepthGather { int32_t c[4]; uint32_t dmax[4]; double weight[4], vnew[4], vold[4]; };
    __device__ __forceinline__ void request(uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4, DepthGather& g) const {
        // the change of every window's depth, windows that coincide folded into the first of them (the if-chains of
        // assgn.rs:259-284 written as sums of comparisons: no divergent paths)
        const int32_t e21 = w2 == w1, e31 = w3 == w1, e41 = w4 == w1;
        const int32_t e32 = w3 == w2, e42 = w4 == w2, e43 = w4 == w3;
        g.c[0] = -1 - e21 + e31 + e41;
        g.c[1] = e21 ? 0 : -1 + e32 + e42;
        g.c[2] = (e31 | e32) ? 0 : 1 + e43;
        g.c[3] = (e41 | e42 | e43) ? 0 : 1;
        const uint32_t w[4] = {w1, w2, w3, w4};
        uint32_t word[4];
#pragma unroll
        for (int i = 0; i < 4; i++) word[i] = wd[w[i]];
        const uint32_t last = V->lut_depth - 1;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t d_old = word[i] & DEPTH_MASK, row = (word[i] >> 25) << V->lut_shift;
            const uint32_t d_new = static_cast<uint32_t>(static_cast<int32_t>(d_old) + g.c[i]);
            g.weight[i] = ww[w[i]];
            g.vnew[i] = V->lut[row + min(d_new, last)];
            g.vold[i] = V->lut[row + min(d_old, last)];
            g.dmax[i] = max(d_new, d_old);
        }
    }
    __device__ __forceinline__ double finish(const DepthGather& g) const {
        uint32_t deepest = 0;
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool live = g.c[i] != 0 && g.weight[i] != 0.0;             // c == 0: no change; weight 0: WindowDistr::TRIVIAL
            deepest = max(deepest, live ? g.dmax[i] : 0u);
            const double term = live ? g.weight[i] * g.vnew[i] - g.weight[i] * g.vold[i] : 0.0;
            sum = i == 0 ? term : sum + term;
        }
        if (deepest > V->lut_depth - 1) atomicMax(V->overflow, 1u);         // every chain of the batch is repeated
        return sum;
    }
    __device__ __forceinline__ double depth_lik_diff(uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4) const {
        DepthGather g;
        request(w1, w2, w3, w4, g);
        return finish(g);
    }
    // depth_lik_diff in two halves of two windows: the current location's pair (w1, w2: depths go down) is the same for every
    // alternative location of a read, so the greedy loop requests it once and each alternative's pair (w3, w4: depths go up) once.
    // Inside a pair the coincidence rule of assgn.rs:259-284 is applied as written (w2 == w1: -2 / 0; w4 == w3: +2 / 0); a window
    // shared BETWEEN the pairs changes both halves — the caller detects that (rare) and takes depth_lik_diff instead.
    __device__ __forceinline__ void request_pair(uint32_t wa, uint32_t wb, int32_t dir, PairGather& g) const {
        const int32_t same = wb == wa;
        g.c[0] = dir * (1 + same); g.c[1] = same ? 0 : dir;
        const uint32_t w[2] = {wa, wb};
        uint32_t word[2];
#pragma unroll
        for (int i = 0; i < 2; i++) word[i] = wd[w[i]];
        const uint32_t last = V->lut_depth - 1;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const uint32_t d_old = word[i] & DEPTH_MASK, row = (word[i] >> 25) << V->lut_shift;
            const uint32_t d_new = static_cast<uint32_t>(static_cast<int32_t>(d_old) + g.c[i]);
            g.weight[i] = ww[w[i]];
            g.vnew[i] = V->lut[row + min(d_new, last)];
            g.vold[i] = V->lut[row + min(d_old, last)];
            g.dmax[i] = max(d_new, d_old);
        }
    }
};

// The same with the window weights in LDS (greedy loop, loci without explicit weights): a window is 6 bytes — depth (23 bit) |
// index into wk << 23 in a word, GC bin | index into wc << 7 in a half-word — and its weight the product of two LDS table
// entries, which is how window_weight_kernel made it. Every 8-byte weight gather moved a 128-byte line out of the L2;
// at 5 000 chains those lines were a quarter of the loop's time.
constexpr uint32_t LW_DEPTH_BITS = 23, LW_DEPTH_MASK = (1u << LW_DEPTH_BITS) - 1u;      // nine bits for a table index
struct ChainLW {
    const SolveView* V;
    uint32_t* wd;               // LDS: depth | wk index << 23
    const uint16_t* wh;         // LDS: GC bin | wc index << 7
    const double* wk; const double* wc;     // LDS
    using DepthGather = Chain::DepthGather;
    __device__ __forceinline__ double weight_of(uint32_t word, uint32_t half) const { return wk[word >> LW_DEPTH_BITS] * wc[half >> 7]; }
    __device__ __forceinline__ double wlp_at(uint32_t w) const {
        const uint32_t word = wd[w], half = wh[w];
        const double weight = weight_of(word, half);
        const uint32_t d = word & LW_DEPTH_MASK;
        if (weight == 0.0) return 0.0;
        if (d >= V->lut_depth) { atomicMax(V->overflow, 1u); return 0.0; }
        return weight * V->lut[(half & 0x7Fu) * V->lut_depth + d];
    }
    __device__ __forceinline__ void request(uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4, DepthGather& g) const {
        const int32_t e21 = w2 == w1, e31 = w3 == w1, e41 = w4 == w1;
        const int32_t e32 = w3 == w2, e42 = w4 == w2, e43 = w4 == w3;
        g.c[0] = -1 - e21 + e31 + e41;
        g.c[1] = e21 ? 0 : -1 + e32 + e42;
        g.c[2] = (e31 | e32) ? 0 : 1 + e43;
        g.c[3] = (e41 | e42 | e43) ? 0 : 1;
        const uint32_t w[4] = {w1, w2, w3, w4};
        uint32_t word[4], half[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { word[i] = wd[w[i]]; half[i] = wh[w[i]]; }
        const uint32_t last = V->lut_depth - 1;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t d_old = word[i] & LW_DEPTH_MASK, row = (half[i] & 0x7Fu) << V->lut_shift;
            const uint32_t d_new = static_cast<uint32_t>(static_cast<int32_t>(d_old) + g.c[i]);
            g.vnew[i] = V->lut[row + min(d_new, last)];
            g.vold[i] = V->lut[row + min(d_old, last)];
            g.dmax[i] = max(d_new, d_old);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) g.weight[i] = weight_of(word[i], half[i]);
    }
    __device__ __forceinline__ double finish(const DepthGather& g) const {
        uint32_t deepest = 0;
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool live = g.c[i] != 0 && g.weight[i] != 0.0;
            deepest = max(deepest, live ? g.dmax[i] : 0u);
            const double term = live ? g.weight[i] * g.vnew[i] - g.weight[i] * g.vold[i] : 0.0;
            sum = i == 0 ? term : sum + term;
        }
        if (deepest > V->lut_depth - 1) atomicMax(V->overflow, 1u);
        return sum;
    }
    __device__ __forceinline__ double depth_lik_diff(uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4) const {
        DepthGather g;
        request(w1, w2, w3, w4, g);
        return finish(g);
    }
    // see Chain::request_pair
    __device__ __forceinline__ void request_pair(uint32_t wa, uint32_t wb, int32_t dir, PairGather& g) const {
        const int32_t same = wb == wa;
        g.c[0] = dir * (1 + same); g.c[1] = same ? 0 : dir;
        const uint32_t w[2] = {wa, wb};
        uint32_t word[2], half[2];
#pragma unroll
        for (int i = 0; i < 2; i++) { word[i] = wd[w[i]]; half[i] = wh[w[i]]; }
        const uint32_t last = V->lut_depth - 1;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const uint32_t d_old = word[i] & LW_DEPTH_MASK, row = (half[i] & 0x7Fu) << V->lut_shift;
            const uint32_t d_new = static_cast<uint32_t>(static_cast<int32_t>(d_old) + g.c[i]);
            g.vnew[i] = V->lut[row + min(d_new, last)];
            g.vold[i] = V->lut[row + min(d_old, last)];
            g.dmax[i] = max(d_new, d_old);
        }
#pragma unroll
        for (int i = 0; i < 2; i++) g.weight[i] = weight_of(word[i], half[i]);
    }
};

// the `cur` word of a record: served by L2 (agent scope), where a wavefront's own stores arrive in program order
__device__ __forceinline__ uint32_t load_rp_cur(const ChainRec* r) {
    return __hip_atomic_load(&r->rp_cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_rp_cur(ChainRec* r, uint32_t v) {
    __hip_atomic_store(&r->rp_cur, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the immutable part of a record: two 16-byte loads
struct RecBody { uint32_t meta; double lp0, lp1; uint32_t win0, win1; };
__device__ __forceinline__ RecBody load_body(const ChainRec* r) {
    const uint4 a = *reinterpret_cast<const uint4*>(r);
    const uint4 b = *(reinterpret_cast<const uint4*>(r) + 1);
    RecBody o;
    o.meta = a.y;
    o.lp0 = __hiloint2double(static_cast<int>(a.w), static_cast<int>(a.z));
    o.lp1 = __hiloint2double(static_cast<int>(b.y), static_cast<int>(b.x));
    o.win0 = b.z; o.win1 = b.w;
    return o;
}
// location t of a record
__device__ __forceinline__ void rec_loc(const RecBody& b, const ExtraLoc* extra, uint32_t t, double* lp, uint32_t* win) {
    if (t == 0) { *lp = b.lp0; *win = b.win0; }
    else if (t == 1) { *lp = b.lp1; *win = b.win1; }
    else { const ExtraLoc e = extra[(b.meta >> 8) + t - 2]; *lp = e.lp; *win = e.win; }
}

// ReassignmentTarget::random (assgn.rs:451-471) from a generator; uniform over the lanes that share `rng`
template <typename CHAIN, typename RNG>
__device__ __forceinline__ void random_move(const CHAIN& C, const RecList& recs, const ExtraLoc* extra, uint32_t nnt, RNG& rng, Move& m) {
    m.slot = static_cast<uint32_t>(rng.below(nnt));
    const uint32_t packed = load_rp_cur(&recs[m.slot]);
    const RecBody b = load_body(&recs[m.slot]);
    const uint32_t rp = packed & 0xFFFFFFu, old_assgn = packed >> 24, total = b.meta & 0xFFu;
    uint32_t new_assgn;
    if (total == 2) new_assgn = 1 - old_assgn;
    else {
        const uint32_t i = 1 + static_cast<uint32_t>(rng.below(total - 1));
        new_assgn = i <= old_assgn ? i - 1 : i;
    }
    m.rp = rp; m.new_assgn = new_assgn;
    uint32_t wo, wn;
    rec_loc(b, extra, old_assgn, &m.lp_old, &wo);
    rec_loc(b, extra, new_assgn, &m.lp_new, &wn);
    m.w1 = wo & 0xFFFFu; m.w2 = wo >> 16; m.w3 = wn & 0xFFFFu; m.w4 = wn >> 16;
    m.ddiff = C.depth_lik_diff(m.w1, m.w2, m.w3, m.w4);
}

// ---- row operations: a row = the LPC lanes of one chain (LPC = 16: DPP inside a row of 16; wider: butterflies) ----
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
// rows of 10 or 12 lanes (six / five chains per wavefront) do not line up with the hardware's rows of 16: their lanes talk
// through the LDS crossbar, a ring inside the row (lane jj + s modulo LPC; spare lanes behind the last row only read)
template <uint32_t LPC>
__device__ __forceinline__ double row_max_f64(double x, uint32_t row_base, uint32_t jj) {
    if constexpr (LPC == 16) {
        x = fmax(x, dpp_f64<0xB1>(x));            // quad_perm [1,0,3,2]
        x = fmax(x, dpp_f64<0x4E>(x));            // quad_perm [2,3,0,1]
        x = fmax(x, dpp_f64<0x141>(x));           // row_half_mirror
        x = fmax(x, dpp_f64<0x140>(x));           // row_mirror
        return x;
    } else if constexpr ((LPC & (LPC - 1)) == 0) {
        for (int o = static_cast<int>(LPC) / 2; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o));
        return x;
    } else {
        // after the shifts 1, 2, 4, 8 a lane has seen the 16 >= LPC lanes that follow it on the ring
#pragma unroll
        for (uint32_t sft = 1; sft < LPC; sft <<= 1) x = fmax(x, __shfl(x, static_cast<int>(row_base + (jj + sft) % LPC)));
        return x;
    }
}
template <uint32_t LPC>
__device__ __forceinline__ double row_sum_f64(double x, uint32_t row_base, uint32_t jj) {
    if constexpr ((LPC & (LPC - 1)) == 0) {
        for (int o = static_cast<int>(LPC) / 2; o > 0; o >>= 1) x += __shfl_xor(x, o);
        return x;
    } else {
        double total = x;                              // outside the main loop: once per chain
        for (uint32_t k = 1; k < LPC; k++) total += __shfl(x, static_cast<int>(row_base + (jj + k) % LPC));
        return total;
    }
}

constexpr uint32_t GREEDY_ROW_TAIL = 4;       // words per row behind the windows (see the kernel)
// LDS of a greedy workgroup for its rows' windows: 4 bytes each; 6 with the weights in LDS, plus the two weight tables (16-byte multiple)
__host__ __device__ inline size_t greedy_lds_windows(uint32_t lpc, uint32_t wstride, uint32_t n_wk, uint32_t n_wc, bool lw) {
    const size_t rows = static_cast<size_t>(64 / lpc) * (lw ? 2 : 1) * wstride;
    const size_t b = lw ? ((rows * 6 + 7) & ~static_cast<size_t>(7)) + static_cast<size_t>(n_wk + n_wc) * 8 : rows * 4;
    return (b + 15) & ~static_cast<size_t>(15);
}

// ---------------- K14 Greedy: 64 / LPC chains per wavefront ----------------
// A candidate read of an iteration as its lane sees it: the record, and the first two of its locations beyond the second
struct GreedyCand { uint32_t pick, rpc; RecBody b; };
struct GreedyExt { double lp2, lp3; uint32_t win2, win3; };
constexpr uint32_t GREEDY_INLINE_LOCS = 4;     // locations of a read the pipelined path holds in registers; reads with more take loads
// selects, not branches: as nested conditionals this became a tree of divergent branches (600 clock ticks per iteration)
__device__ __forceinline__ void cand_loc(const RecBody& b, const GreedyExt& e, uint32_t t, double* lp, uint32_t* win) {
    double l = b.lp0; uint32_t w = b.win0;
    l = t == 1 ? b.lp1 : l; w = t == 1 ? b.win1 : w;
    l = t == 2 ? e.lp2 : l; w = t == 2 ? e.win2 : w;
    l = t >= 3 ? e.lp3 : l; w = t >= 3 ? e.win3 : w;
    *lp = l; *win = w;
}

// With the weights in LDS a workgroup is two wavefronts that share the two tables (nothing else: after one barrier they run apart)
template <uint32_t LPC, bool LW, uint32_t GREEDY_FORM = 0>
__global__ __launch_bounds__(LW ? 128 : 64) void greedy_loop_kernel(const SolveView V, const uint32_t n_chains) {
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ uint32_t flagged;
    constexpr uint32_t CPW = 64 / LPC, WAVES = LW ? 2u : 1u, ROWS = CPW * WAVES;
    using ChainT = typename std::conditional<LW, ChainLW, Chain>::type;
    // a batch whose initialisation raised a flag (a chain's run of further locations was too short, ...) is repeated by the host:
    // its records are incomplete and must not be followed
    if (threadIdx.x == 0) flagged = __hip_atomic_load(V.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (flagged != 0u) return;
    if (V.prio_mode == 1) __builtin_amdgcn_s_setprio(3);
    const uint32_t W = V.wstride;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // lanes behind the last row (LPC 10, 12: lanes 60..63) ride along with it as lanes without a candidate
    const uint32_t row = min(lane / LPC, CPW - 1), row_base = row * LPC, jj = lane - row_base;
    const uint32_t wg_row = wave * CPW + row;
    const uint32_t chain_raw = blockIdx.x * ROWS + wg_row;
    const bool live_row = chain_raw < n_chains;
    const uint32_t chain = live_row ? chain_raw : n_chains - 1;                  // a spare row shadows the last chain, without effects
    uint32_t* wd = reinterpret_cast<uint32_t*>(smem) + static_cast<size_t>(wg_row) * W;
    const uint32_t gi = chain / V.attempts;
    // the parts of the chain's list of non-trivial reads (RecList): four words per row in LDS, behind everything else
    // per row, behind the windows: [0..3] the parts of the record list
    uint32_t* row_cum = reinterpret_cast<uint32_t*>(smem + greedy_lds_windows(LPC, V.wstride, V.n_wk, V.n_wc, LW)) + static_cast<size_t>(wg_row) * GREEDY_ROW_TAIL;
    if (jj < 4) row_cum[jj] = V.c_seg[static_cast<uint64_t>(chain) * 4 + jj];
    RecList recs{V.recs + static_cast<uint64_t>(chain) * V.rstride, row_cum, V.seg_reads};
    {
        // the bounds of the list's parts in registers of the lane (three LDS reads per pick less: 312 -> 309 ms)
        const uint32_t* cg = V.c_seg + static_cast<uint64_t>(chain) * 4;
        recs.r1 = cg[1]; recs.r2 = cg[2]; recs.r3 = cg[3]; recs.in_regs = true;
    }
    const ExtraLoc* extra = V.extra + static_cast<uint64_t>(chain) * V.extra_cap;
    const uint32_t total_w = V.c_totw[chain];
    const uint8_t* ggc = V.c_gc + static_cast<uint64_t>(chain) * W;
    const uint32_t* gd = V.c_depth + static_cast<uint64_t>(chain) * W;
    ChainT C;
    if constexpr (LW) {
        // [ROWS][W] words, [ROWS][W] half-words, then the two weight tables
        uint16_t* wh = reinterpret_cast<uint16_t*>(smem + static_cast<size_t>(ROWS) * W * 4) + static_cast<size_t>(wg_row) * W;
        double* lwk = reinterpret_cast<double*>(smem + ((static_cast<size_t>(ROWS) * W * 6 + 7) & ~static_cast<size_t>(7)));
        double* lwc = lwk + V.n_wk;
        const uint32_t* guc = V.c_uc + static_cast<uint64_t>(chain) * W;
        for (uint32_t w = jj; w < total_w && jj < LPC; w += LPC) {
            const uint32_t uc = guc[w];
            wd[w] = gd[w] | ((uc & 0xFFFFu) << LW_DEPTH_BITS);
            wh[w] = static_cast<uint16_t>(ggc[w] | ((uc >> 16) << 7));
        }
        for (uint32_t i = threadIdx.x; i < V.n_wk; i += 64 * WAVES) lwk[i] = V.wk[i];
        for (uint32_t i = threadIdx.x; i < V.n_wc; i += 64 * WAVES) lwc[i] = V.wc[i];
        C = ChainLW{&V, wd, wh, lwk, lwc};
        __syncthreads();                                                         // the tables; from here on the wavefronts run apart
    } else {
        for (uint32_t w = jj; w < total_w && jj < LPC; w += LPC) wd[w] = gd[w] | (static_cast<uint32_t>(ggc[w]) << 25);
        C = Chain{&V, wd, V.c_ww + static_cast<uint64_t>(chain) * W};
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t nnt = V.c_nnt[chain];
    // depth_lik = sum over windows (recalc_likelihood, assgn.rs:347-350)
    double depth_lik = 0.0;
    for (uint32_t w = jj; w < total_w && jj < LPC; w += LPC) depth_lik += C.wlp_at(w);
    depth_lik = row_sum_f64<LPC>(depth_lik, row_base, jj);
    double aln_lik = V.c_aln[chain];
    Xoshiro rng; rng.seed(V.seeds[chain]);
    uint64_t n_iter = 0, n_acc = 0;
    const double rel_contrib = V.depth_contrib / V.aln_contrib;
    double depth_mine = 0.0, aln_mine = 0.0;                                     // likelihood changes of the moves this lane applied

    bool done = nnt == 0 || !live_row;
    uint64_t phase[5] = {0, 0, 0, 0, 0}, sub[4] = {0, 0, 0, 0}, t_loop0 = 0;
    if (__any(!done)) {
        const uint32_t nnt1 = max(nnt, 1u);
        const uint64_t max_iter = max(static_cast<uint64_t>(100000), static_cast<uint64_t>(V.solver.plato_size) * 100);
        // max_abs_random (stoch.rs:19-22) over INIT_ITER = 100 random targets, the same in every lane of the row
        double max_abs = 0.0;
        for (uint32_t i = 0; i < 100; i++) {
            Move m;
            if (!done) {
                random_move(C, recs, extra, nnt1, rng, m);
                max_abs = fmax(max_abs, fabs(V.depth_contrib * m.ddiff + V.aln_contrib * (m.lp_new - m.lp_old)));
            }
        }
        const double min_diff = fmax(1e-10 * max_abs, 1e-14);                 // minimum_allowed_diff (stoch.rs:27-29)
        // Greedy::solve_nontrivial (stoch.rs:81-120). Lane jj of the row evaluates the jj-th read of the iteration's sample.
        //
        // The random stream does not depend on what the moves do, so the records an iteration will look at are known ahead: the
        // loop is a software pipeline over three record slots. Iteration i, in program order:
        //   1. its candidates are in registers (record of slot i, its further locations): depths from LDS, then ISSUE the
        //      gathers of the table entries and window weights (twelve per alternative location);
        //   2. ISSUE the further locations of iteration i + 1's candidates (their records arrived an iteration ago);
        //   3. draw the sample of iteration i + 3 and ISSUE its record loads;
        //   4. wait for the gathers of step 1 only — the vector-memory counter is in order, everything issued in 2 and 3 is
        //      younger and stays in flight — score, pick the row's best, apply the move.
        // A record therefore has three iterations to come from HBM, and an iteration waits for one L2 round trip.
        // A move changes the current location of one read; the records of the next three iterations were requested before
        // it: the last three moves are kept and applied to a record when it is used.
        const uint32_t S = min(V.solver.sample_size, nnt1);
        const bool cand = jj < S;
        auto sample = [&]() -> uint32_t {
            // non_trivial_reads.sample(rng, S): one draw of the chain's generator as a key, the S picks as counter draws under it,
            // repeats skipped (our adaptor, oracle/lcty_oracle.h). A sample almost never repeats an index (S^2 / 2 nnt)
            const uint64_t key = rng.next();
            uint32_t idx = cand ? static_cast<uint32_t>(__umul64hi(counter_u64(key, jj), static_cast<uint64_t>(nnt1))) : 0xFFFFFFFFu - jj;
            bool dup = false;
            if constexpr (LPC == 16) {
                // every unordered pair of the row meets in one of eight rotations
                const int v = static_cast<int>(idx);
                dup |= __builtin_amdgcn_update_dpp(0, v, 0x121, 0xF, 0xF, false) == v;
                dup |= __builtin_amdgcn_update_dpp(0, v, 0x122, 0xF, 0xF, false) == v;
                dup |= __builtin_amdgcn_update_dpp(0, v, 0x123, 0xF, 0xF, false) == v;
                dup |= __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, false) == v;
                dup |= __builtin_amdgcn_update_dpp(0, v, 0x125, 0xF, 0xF, false) == v;
                dup |= __builtin_amdgcn_update_dpp(0, v, 0x126, 0xF, 0xF, false) == v;
                dup |= __builtin_amdgcn_update_dpp(0, v, 0x127, 0xF, 0xF, false) == v;
                dup |= __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, false) == v;
            } else {
                for (uint32_t d = 1; d < LPC; d++) {
                    const uint32_t other = static_cast<uint32_t>(__shfl(static_cast<int>(idx), static_cast<int>(row_base + ((jj + d) % LPC))));
                    dup |= other == idx;
                }
            }
            unsigned long long dup_rows = __ballot(dup && cand);
            while (dup_rows) {
                // the picks of that row again, in order, repeats skipped, as the rule says
                const uint32_t r = static_cast<uint32_t>(__ffsll(static_cast<long long>(dup_rows)) - 1) / LPC;
                const uint64_t kb = uniform64(__shfl(key, static_cast<int>(r * LPC)));
                const uint32_t nb = __builtin_amdgcn_readlane(nnt1, r * LPC);
                const uint32_t Sb = min(V.solver.sample_size, nb);
                uint64_t ctr = 0;
                for (uint32_t j = 0; j < Sb; j++) {
                    uint32_t pick;
                    bool again;
                    do {
                        pick = static_cast<uint32_t>(__umul64hi(counter_u64(kb, ctr++), static_cast<uint64_t>(nb)));
                        again = __ballot(row == r && jj < j && idx == pick) != 0ull;
                    } while (again);
                    if (row == r && jj == j) idx = pick;
                }
                const unsigned long long row_lanes = (LPC == 64 ? ~0ull : ((1ull << LPC) - 1ull)) << (r * LPC);
                dup_rows &= ~row_lanes;
            }
            return idx;
        };
        // Loads that stay in flight across iterations are issued field by field (relaxed atomic loads: the compiler neither
        // merges nor splits them). A merged 12- or 16-byte load comes back as a register tuple, and a tuple whose parts live
        // on for different lengths gets copied apart by the register allocator right after the load — which waits for it.
        auto field32 = [](const void* p, uint32_t byte_off) -> uint32_t {
            return __hip_atomic_load(reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(p) + byte_off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        };
        auto field64 = [](const void* p, uint32_t byte_off) -> double {
            return __longlong_as_double(static_cast<long long>(__hip_atomic_load(
                reinterpret_cast<const unsigned long long*>(static_cast<const uint8_t*>(p) + byte_off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT)));
        };
        auto request_record = [&](GreedyCand& c) {
            c.pick = sample();
            const ChainRec* r = &recs[cand ? c.pick : 0u];
            c.rpc = load_rp_cur(r);
            c.b.meta = field32(r, 4); c.b.lp0 = field64(r, 8); c.b.lp1 = field64(r, 16); c.b.win0 = field32(r, 24); c.b.win1 = field32(r, 28);
        };
        // the first two further locations of a record that has arrived (reads with two locations: the chain's first entry, unused)
        auto request_ext = [&](const GreedyCand& c, GreedyExt& e) {
            const uint32_t nloc = c.b.meta & 0xFFu;
            const ExtraLoc* p = extra + (cand && nloc > 2 ? (c.b.meta >> 8) : 0u);   // the run has spare entries behind it
            e.lp2 = field64(p, 0); e.win2 = field32(p, 8); e.lp3 = field64(p, 16); e.win3 = field32(p, 24);
        };
        uint32_t curr_plato = 0;
        uint64_t iter = 0;
        // the last three moves of the row, newest first (slot 0xFFFFFFFF: none)
        uint32_t h1s = 0xFFFFFFFFu, h1t = 0, h2s = 0xFFFFFFFFu, h2t = 0, h3s = 0xFFFFFFFFu, h3t = 0;

        // one iteration: A = its candidates (arrived), EA their further locations (arrived); N = the candidates of the next
        // iteration (arrived), EN receives their further locations; F = the slot the sample three iterations ahead goes to (= A's)
        auto iteration = [&](auto slot_tag, GreedyCand& A, const GreedyExt& EA, const GreedyCand& N, GreedyExt& EN) {
            constexpr uint32_t SLOT = decltype(slot_tag)::value;
            constexpr bool TIMED = (GREEDY_FORM & 32u) != 0;
            uint64_t tk[6] = {0, 0, 0, 0, 0, 0}, ts[3] = {0, 0, 0}, tw0 = 0;
            auto subtick = [&](int k) { if constexpr (TIMED) { __builtin_amdgcn_sched_barrier(0); ts[k] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } };
            auto tick = [&](int k) { if constexpr (TIMED) { __builtin_amdgcn_sched_barrier(0); tk[k] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } };
            tick(0);
            if constexpr (TIMED) {                     // how long the further locations requested an iteration ago are still on their way
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0); tw0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
            }
            const RecBody b = A.b;
            const uint32_t pick = A.pick, rpc0 = A.rpc;
            const uint32_t nloc = b.meta & 0xFFu;
            uint32_t cur = rpc0 >> 24;
            cur = pick == h3s ? h3t : cur; cur = pick == h2s ? h2t : cur; cur = pick == h1s ? h1t : cur;
            // lanes without a candidate (beyond the sample, rows that are done or spare) look at window 0 four times: no change, no effect
            const uint32_t n_alt = (cand && !done) ? nloc - 1 : 0u;
            const bool deep = n_alt != 0 && nloc > GREEDY_INLINE_LOCS;           // some locations of this read are not in registers
            double cur_lp = 0.0; uint32_t cur_w = 0;
            if (n_alt) cand_loc(b, EA, min(cur, GREEDY_INLINE_LOCS - 1), &cur_lp, &cur_w);
            if (__any(deep && cur >= GREEDY_INLINE_LOCS)) {                      // rare: the current location itself is beyond the fourth
                if (deep && cur >= GREEDY_INLINE_LOCS) rec_loc(b, extra, cur, &cur_lp, &cur_w);
                asm volatile("" : "+v"(cur_lp), "+v"(cur_w));                     // the wait for this load stays inside the rare branch
            }
            const uint32_t w1 = cur_w & 0xFFFFu, w2 = cur_w >> 16;
            if constexpr (TIMED) asm volatile("" :: "v"(w1), "v"(w2), "v"(cur_lp));
            subtick(0);
            // up to three alternatives per lane, all requested before any is used. The pair of the current location is the same
            // for all of them: requested once; an alternative adds the pair of its own two windows.
            constexpr uint32_t NA = GREEDY_INLINE_LOCS - 1;
            double lp_t[NA]; uint32_t win_t[NA], t_of[NA];
            PairGather gc, ga[NA];
            bool cross[NA];
            C.request_pair(w1, w2, -1, gc);
            subtick(1);
#pragma unroll
            for (uint32_t u = 0; u < NA; u++) {
                if (u == 1) subtick(2);
                const bool has = u < n_alt && !(deep && u + (u >= cur ? 1u : 0u) >= GREEDY_INLINE_LOCS);
                t_of[u] = u + (u >= cur ? 1u : 0u);
                lp_t[u] = 0.0; win_t[u] = 0; cross[u] = false;
                if (u == 0 || __any(has)) {
                    if (has) cand_loc(b, EA, t_of[u], &lp_t[u], &win_t[u]);
                    const uint32_t a3 = has ? (win_t[u] & 0xFFFFu) : 0u, a4 = has ? (win_t[u] >> 16) : 0u;
                    C.request_pair(a3, a4, 1, ga[u]);
                    // a window shared between the two pairs (windows 0 and 1 carry no distribution: sharing them changes nothing)
                    cross[u] = has && ((a3 > 1 && (a3 == w1 || a3 == w2)) || (a4 > 1 && (a4 == w1 || a4 == w2)));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            tick(1);
            request_ext(N, EN);
            __builtin_amdgcn_sched_barrier(0);
            request_record(A);                                                  // A's registers have been read: the slot takes the sample of i + 3
            __builtin_amdgcn_sched_barrier(0);
            tick(2);
            // best_read_improvement (assgn.rs:287-317): the alternatives in order, a later one only when strictly better
            double best_improv = -INFINITY, lp_new = 0.0, ddiff = 0.0;
            uint32_t new_assgn = 0, w3 = 0, w4 = 0;
            uint32_t deepest = 0;
            const double t1 = pair_term(gc, 0, &deepest);
            const double t12 = t1 + pair_term(gc, 1, &deepest);
#pragma unroll
            for (uint32_t u = 0; u < NA; u++) {
                const bool has = u < n_alt && !(deep && t_of[u] >= GREEDY_INLINE_LOCS);
                if (u == 0 || __any(has)) {
                    const double t3 = pair_term(ga[u], 0, &deepest);
                    double dd = (t12 + t3) + pair_term(ga[u], 1, &deepest);
                    if (__any(cross[u])) {
                        if (cross[u]) dd = C.depth_lik_diff(w1, w2, win_t[u] & 0xFFFFu, win_t[u] >> 16);
                    }
                    const double improv = lp_t[u] + rel_contrib * dd;
                    if (has && (u == 0 || improv > best_improv)) {
                        best_improv = improv; new_assgn = t_of[u]; w3 = win_t[u] & 0xFFFFu; w4 = win_t[u] >> 16; lp_new = lp_t[u]; ddiff = dd;
                    }
                }
            }
            if (deepest > V.lut_depth - 1) atomicMax(V.overflow, 1u);            // every chain of the batch is repeated
            if (__any(deep)) {
                // reads with more than four locations: the ones beyond the fourth from the chain's run, one at a time, in their place
                // in the order (a read's alternatives are visited by ascending location; those in registers may come after these
                // only when the current location is beyond the fourth — then none of the inline ones was skipped)
                for (uint32_t u = 0; __any(deep && u < n_alt); u++) {
                    const uint32_t t = u + (u >= cur ? 1u : 0u);
                    if (deep && u < n_alt && (u >= NA || t >= GREEDY_INLINE_LOCS)) {
                        double lp_u; uint32_t win_u;
                        rec_loc(b, extra, t, &lp_u, &win_u);
                        const uint32_t a3 = win_u & 0xFFFFu, a4 = win_u >> 16;
                        const double dd = C.depth_lik_diff(w1, w2, a3, a4);
                        const double improv = lp_u + rel_contrib * dd;
                        // alternatives arrive in ascending u here as in the unrolled part: the inline ones have u < NA and t < 4
                        if (improv > best_improv) { best_improv = improv; new_assgn = t; w3 = a3; w4 = a4; lp_new = lp_u; ddiff = dd; }
                    }
                }
            }
            const double my_improv = n_alt ? V.aln_contrib * (best_improv - cur_lp) : -INFINITY;
            if constexpr (TIMED) asm volatile("" :: "v"(my_improv));
            tick(3);
            // first candidate (sample order) with the largest improvement above min_diff (stoch.rs:103-109)
            const double best = row_max_f64<LPC>(my_improv, row_base, jj);
            const unsigned long long who = __ballot(n_alt && my_improv == best);
            const unsigned long long who_row = (who >> row_base) & (LPC == 64 ? ~0ull : ((1ull << (LPC & 63u)) - 1ull));
            const uint32_t src = who_row ? static_cast<uint32_t>(__ffsll(static_cast<long long>(who_row))) - 1u : 0u;
            const bool moved = !done && who_row != 0ull && best > min_diff;
            if constexpr (TIMED) asm volatile("" :: "s"(who));
            tick(4);
            // reassign (assgn.rs:331-343) by the lane that holds the move; the others only learn which list slot changed.
            // The lane's share of the likelihood is added up at the end.
            if (moved && jj == src) {
                atomicAdd(&wd[w3], 1u); atomicAdd(&wd[w4], 1u);                   // the depth field never borrows from the GC bits
                atomicSub(&wd[w1], 1u); atomicSub(&wd[w2], 1u);
                store_rp_cur(&recs[pick], (rpc0 & 0xFFFFFFu) | (new_assgn << 24));
                depth_mine += ddiff;
                aln_mine += lp_new - cur_lp;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t moved_slot = static_cast<uint32_t>(__shfl(static_cast<int>(pick), static_cast<int>(row_base + src)));
            const uint32_t moved_to = static_cast<uint32_t>(__shfl(static_cast<int>(new_assgn), static_cast<int>(row_base + src)));
            h3s = h2s; h3t = h2t; h2s = h1s; h2t = h1t;
            h1s = moved ? moved_slot : 0xFFFFFFFFu; h1t = moved_to;
            if constexpr (TIMED) {
                asm volatile("" :: "v"(h1s), "v"(h1t));
                tick(5);
                for (int k = 0; k < 5; k++) phase[k] += tk[k + 1] - tk[k];
                sub[0] += ts[0] - tw0; sub[1] += ts[1] - ts[0]; sub[2] += ts[2] - ts[1]; sub[3] += tw0 - tk[0];
            }
            if (!done) {
                n_iter++; iter++;
                if (moved) { n_acc++; curr_plato = 0; }
                else {
                    curr_plato++;
                    if (curr_plato > V.solver.plato_size) done = true;
                }
                if (iter >= max_iter) done = true;
            }
        };
        GreedyCand R0, R1, R2;
        GreedyExt E0, E1, E2;
        if constexpr ((GREEDY_FORM & 32u) != 0) t_loop0 = __builtin_amdgcn_s_memtime();
        request_record(R0); request_record(R1); request_record(R2);
        request_ext(R0, E0);
        while (__any(!done)) {
            iteration(std::integral_constant<uint32_t, 0>{}, R0, E0, R1, E1);
            if (!__any(!done)) break;
            iteration(std::integral_constant<uint32_t, 1>{}, R1, E1, R2, E2);
            if (!__any(!done)) break;
            iteration(std::integral_constant<uint32_t, 2>{}, R2, E2, R0, E0);
        }
    }
    if constexpr ((GREEDY_FORM & 32u) != 0) {
        if (lane == 0 && V.dbg) {
            double* d = V.dbg + (static_cast<size_t>(blockIdx.x) * WAVES + wave) * 12;
            for (int k = 0; k < 5; k++) d[k] = static_cast<double>(phase[k]);
            d[5] = static_cast<double>(__builtin_amdgcn_s_memtime() - t_loop0); d[6] = static_cast<double>(n_iter); d[7] = 0.0;
            for (int k = 0; k < 4; k++) d[8 + k] = static_cast<double>(sub[k]);
        }
    }
    depth_mine = row_sum_f64<LPC>(depth_mine, row_base, jj); aln_mine = row_sum_f64<LPC>(aln_mine, row_base, jj);
    depth_lik += depth_mine; aln_lik += aln_mine;
    if (jj == 0 && live_row) {
        const double lik = V.depth_contrib * depth_lik + V.aln_contrib * aln_lik;       // assgn.rs:235-237
        V.liks[chain] = (V.priors ? V.priors[gi] : 0.0) + lik;                          // solve.rs:827
        V.parts[4 * chain] = aln_lik; V.parts[4 * chain + 1] = depth_lik; V.parts[4 * chain + 2] = static_cast<double>(n_iter);
        V.parts[4 * chain + 3] = static_cast<double>(n_acc);
    }
}

// ---------------- K14 SimAnneal: one chain wavefront + one staging wavefront ----------------
constexpr uint32_t RING = 64;                  // staged positions of the random stream (power of two)
constexpr uint32_t SPIN_LIMIT = 200u * 1000u * 1000u;    // bounded waits: a lost hand-shake becomes an error, not a hang
// one staged draw: the read it would pick and everything about that read that no move can change — its possible
// locations on the genotype in order (ln-probability, tweaked windows w1 | w2 << 16); nloc > 4: not staged, read from the record
struct __attribute__((aligned(16))) StagedRead {
    uint64_t draw;
    uint32_t rp, nloc;
    double lp[4];
    uint32_t win[4];
};
struct AnnealRing {
    StagedRead pos[RING];
    uint64_t rng[4];
    uint32_t produced, consumed, stop, go;
    uint32_t cum[4];            // the parts of the chain's record list (RecList)
};

// Where the chain's window weights are (MODE): 0 gathered from its row in L2 with the table entries (the same latency chain, the
// least LDS); 1 in LDS as they are (8 bytes per window); 2 as the greedy loop has them: the two table indices packed next to depth
// and GC bin (6 bytes per window) and the two weight tables in LDS — no weight gathers and 15 KB per chain. Next to the greedy
// chains of the following locus the LDS of the device is what runs out, and every gather shares the L2 -> L1 path with theirs.
template <int MODE>
__global__ __launch_bounds__(128, 4) void anneal_loop_kernel(const SolveView V) {
    constexpr bool WWL = MODE == 1, LW = MODE == 2;
    using ChainT = typename std::conditional<LW, ChainLW, Chain>::type;
    extern __shared__ __align__(32) uint8_t smem[];
    __shared__ uint32_t flagged;
    if (threadIdx.x == 0) flagged = __hip_atomic_load(V.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (flagged != 0u) return;                     // the batch is repeated by the host (see greedy_loop_kernel)
    // These are few wavefronts with one dependent chain of instructions each; in a queue of loci they share their SIMDs with the
    // greedy wavefronts of the next locus, which always have an instruction ready. The issue arbiter takes the higher priority first.
    if (V.prio_mode == 0) __builtin_amdgcn_s_setprio(3);
    const uint32_t W = V.wstride;
    // [W] window weights first (MODE 1) or the two weight tables (MODE 2), then [W] depth words, [W] half-words (MODE 2), then the
    // ring the second wavefront fills
    const size_t head_bytes = WWL ? static_cast<size_t>(W) * 8 : LW ? static_cast<size_t>(V.n_wk + V.n_wc) * 8 : 0;
    const size_t half_bytes = LW ? ((static_cast<size_t>(W) * 2 + 31) & ~static_cast<size_t>(31)) : 0;
    double* lww = reinterpret_cast<double*>(smem);
    uint32_t* wd = reinterpret_cast<uint32_t*>(smem + head_bytes);
    uint16_t* wh = reinterpret_cast<uint16_t*>(smem + head_bytes + ((static_cast<size_t>(W) * 4 + 31) & ~static_cast<size_t>(31)));
    AnnealRing* ring = reinterpret_cast<AnnealRing*>(smem + head_bytes + ((static_cast<size_t>(W) * 4 + 31) & ~static_cast<size_t>(31)) + half_bytes);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t chain = blockIdx.x;
    const uint32_t gi = chain / V.attempts;
    const uint64_t seed = uniform64(V.seeds[chain]);
    const RecList recs{V.recs + static_cast<uint64_t>(chain) * V.rstride, ring->cum, V.seg_reads};
    if (threadIdx.x < 4) ring->cum[threadIdx.x] = V.c_seg[static_cast<uint64_t>(chain) * 4 + threadIdx.x];
    const ExtraLoc* extra = V.extra + static_cast<uint64_t>(chain) * V.extra_cap;
    const uint32_t total_w = V.c_totw[chain];
    const double* gww = V.c_ww + static_cast<uint64_t>(chain) * W;
    {
        const uint8_t* ggc = V.c_gc + static_cast<uint64_t>(chain) * W;
        const uint32_t* gd = V.c_depth + static_cast<uint64_t>(chain) * W;
        if constexpr (LW) {
            const uint32_t* guc = V.c_uc + static_cast<uint64_t>(chain) * W;
            for (uint32_t w = lane; wave == 0 && w < total_w; w += 64) {
                const uint32_t uc = guc[w];
                wd[w] = gd[w] | ((uc & 0xFFFFu) << LW_DEPTH_BITS);
                wh[w] = static_cast<uint16_t>(ggc[w] | ((uc >> 16) << 7));
            }
            for (uint32_t i = threadIdx.x; i < V.n_wk; i += 128) lww[i] = V.wk[i];
            for (uint32_t i = threadIdx.x; i < V.n_wc; i += 128) lww[V.n_wk + i] = V.wc[i];
        } else {
            for (uint32_t w = lane; wave == 0 && w < total_w; w += 64) {
                wd[w] = gd[w] | (static_cast<uint32_t>(ggc[w]) << 25);
                if (WWL) lww[w] = gww[w];
            }
        }
    }
    const uint32_t nnt = V.c_nnt[chain];
    if (wave == 0 && lane == 0) { ring->produced = 0; ring->consumed = 0; ring->stop = 0; ring->go = 0; }
    __syncthreads();
    if (wave == 1) {
        // ---- producer: runs the random stream ahead of the chain and stages, for every draw taken as "the read of a
        // move", that read's record (one 32-byte gather; nothing of it but the current location depends on the chain's state)
        __syncthreads();                                                     // hand-over of the stream (below)
        if (__hip_atomic_load(&ring->go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
            Xoshiro prng;
            prng.s0 = uniform64(ring->rng[0]); prng.s1 = uniform64(ring->rng[1]);
            prng.s2 = uniform64(ring->rng[2]); prng.s3 = uniform64(ring->rng[3]);
            uint32_t produced = 0, idle = 0;
            for (;;) {
                if (__hip_atomic_load(&ring->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                const uint32_t c = __hip_atomic_load(&ring->consumed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t n = RING - (produced - c);
                if (n == 0) {
                    __builtin_amdgcn_s_sleep(4);
                    if (++idle > SPIN_LIMIT) { atomicMax(V.overflow, 3u); break; }
                    continue;
                }
                idle = 0;
                uint64_t mine = 0;
                for (uint32_t kk = 0; kk < n; kk++) {
                    const uint64_t v = prng.next();
                    if (lane == kk) mine = v;
                }
                if (lane < n) {
                    const uint32_t slot = static_cast<uint32_t>(__umul64hi(mine, static_cast<uint64_t>(nnt)));
                    const uint32_t rp = recs[slot].rp_cur & 0xFFFFFFu;
                    const RecBody b = load_body(&recs[slot]);
                    StagedRead* e = &ring->pos[(produced + lane) & (RING - 1)];
                    const uint32_t nloc = b.meta & 0xFFu;
                    e->draw = mine; e->rp = rp; e->nloc = nloc;
                    e->lp[0] = b.lp0; e->lp[1] = b.lp1; e->win[0] = b.win0; e->win[1] = b.win1;
                    if (nloc > 2 && nloc <= 4) {
                        for (uint32_t t = 2; t < nloc; t++) { const ExtraLoc x = extra[(b.meta >> 8) + t - 2]; e->lp[t] = x.lp; e->win[t] = x.win; }
                    }
                }
                produced += n;
                __hip_atomic_store(&ring->produced, produced, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        return;
    }
    ChainT C;
    if constexpr (LW) C = ChainLW{&V, wd, wh, lww, lww + V.n_wk};
    else C = Chain{&V, wd, WWL ? lww : gww};
    // depth_lik = sum over windows (recalc_likelihood, assgn.rs:347-350)
    double depth_lik = 0.0;
    for (uint32_t w = lane; w < total_w; w += 64) depth_lik += C.wlp_at(w);
    for (int o2 = 32; o2 > 0; o2 >>= 1) depth_lik += __shfl_xor(depth_lik, o2);
    double aln_lik = V.c_aln[chain];
    Xoshiro rng; rng.seed(seed);
    uint64_t n_iter = 0, n_acc = 0;
    bool handed_over = false;

    auto improvement = [&](const Move& m) -> double {                         // calculate_improvement (assgn.rs:321-328)
        return V.depth_contrib * m.ddiff + V.aln_contrib * (m.lp_new - m.lp_old);
    };
    auto reassign = [&](const Move& m) {                                      // assgn.rs:331-343 (wave-uniform move)
        n_acc++;
        depth_lik += m.ddiff;                                                 // nothing moved since it was evaluated
        aln_lik += m.lp_new - m.lp_old;
        if (lane == 0) {
            // LDS atomics (no return value): four operations in program order without a read-modify-write round trip each
            atomicAdd(&wd[m.w3], 1u); atomicAdd(&wd[m.w4], 1u);                // the depth field never borrows from the GC bits
            atomicSub(&wd[m.w1], 1u); atomicSub(&wd[m.w2], 1u);
            store_rp_cur(&recs[m.slot], m.rp | (m.new_assgn << 24));
        }
        // one wavefront owns the chain: LDS and vector-memory operations of a wavefront execute in program order, so the
        // following reads see these updates without waiting for the store to complete (no s_waitcnt vmcnt(0) here)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    if (nnt > 0) {
        const uint64_t max_iter = max(static_cast<uint64_t>(100000), static_cast<uint64_t>(V.solver.plato_size) * 100);
        // SimAnneal::solve_nontrivial (stoch.rs:195-245).
        // Every draw of the chain's random stream comes out of an LDS ring that the second wavefront of the workgroup
        // fills ahead of time: it runs the same stream, treats every draw as if it picked the read of a move, and
        // stages that read's record. What is left here per move: the record's current location (L2-warm), two of the
        // staged locations, depth_lik_diff.
        if (lane == 0) { ring->rng[0] = rng.s0; ring->rng[1] = rng.s1; ring->rng[2] = rng.s2; ring->rng[3] = rng.s3; }
        __hip_atomic_store(&ring->go, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        handed_over = true;
        __syncthreads();
        uint32_t consumed = 0;
        bool lost = false;
        auto ring_wait = [&](uint32_t need) -> uint32_t {                  // staged positions (>= need), 0 = hand-shake lost
            uint32_t idle = 0;
            for (;;) {
                const uint32_t avail = __hip_atomic_load(&ring->produced, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - consumed;
                if (avail >= need) return avail;
                __builtin_amdgcn_s_sleep(2);
                if (++idle > SPIN_LIMIT) { lost = true; atomicMax(V.overflow, 3u); return 0; }
            }
        };
        auto ring_f64 = [&](uint32_t off) -> double {                      // rng.random::<f64>() at stream position consumed + off
            return static_cast<double>(ring->pos[(consumed + off) & (RING - 1)].draw >> 11) * (1.0 / 9007199254740992.0);
        };
        // ReassignmentTarget::random (assgn.rs:451-471) at stream position consumed + off, given the `cur` word of its read's record;
        // everything of the move but its depth term; returns the draws it takes
        auto move_pre = [&](uint32_t off, uint32_t packed, Move& m) -> uint32_t {
            const StagedRead* e = &ring->pos[(consumed + off) & (RING - 1)];
            const uint64_t draw = e->draw, draw_next = ring->pos[(consumed + off + 1) & (RING - 1)].draw;
            const uint32_t nloc = e->nloc;
            m.slot = static_cast<uint32_t>(__umul64hi(draw, static_cast<uint64_t>(nnt)));
            const uint32_t rp = packed & 0xFFFFFFu, old_assgn = packed >> 24;
            uint32_t new_assgn;
            if (nloc == 2) new_assgn = 1 - old_assgn;
            else {
                const uint32_t i = 1 + static_cast<uint32_t>(__umul64hi(draw_next, static_cast<uint64_t>(nloc - 1)));
                new_assgn = i <= old_assgn ? i - 1 : i;
            }
            m.rp = rp; m.new_assgn = new_assgn;
            uint32_t w_o, w_n;
            if (nloc <= 4) {
                double lp_o = e->lp[0], lp_n = lp_o; w_o = e->win[0]; w_n = w_o;
#pragma unroll
                for (uint32_t t = 1; t < 4; t++) {
                    const double lp_t = e->lp[t]; const uint32_t w_t = e->win[t];
                    if (t == old_assgn) { lp_o = lp_t; w_o = w_t; }
                    if (t == new_assgn) { lp_n = lp_t; w_n = w_t; }
                }
                m.lp_old = lp_o; m.lp_new = lp_n;
            } else {
                const RecBody b = load_body(&recs[m.slot]);
                rec_loc(b, extra, old_assgn, &m.lp_old, &w_o);
                rec_loc(b, extra, new_assgn, &m.lp_new, &w_n);
                asm volatile("" : "+v"(w_o), "+v"(w_n));                     // the wait for these loads stays in this rare branch
            }
            m.w1 = w_o & 0xFFFFu; m.w2 = w_o >> 16; m.w3 = w_n & 0xFFFFu; m.w4 = w_n >> 16;
            return nloc > 2 ? 2u : 1u;
        };
        auto slot_at = [&](uint32_t off) -> uint32_t {
            return static_cast<uint32_t>(__umul64hi(ring->pos[(consumed + off) & (RING - 1)].draw, static_cast<uint64_t>(nnt)));
        };
        auto ring_move = [&](uint32_t off, Move& m) -> uint32_t {          // with its own load of the `cur` word and its depth term at once
            const uint32_t c = move_pre(off, load_rp_cur(&recs[slot_at(off)]), m);
            m.ddiff = C.depth_lik_diff(m.w1, m.w2, m.w3, m.w4);
            return c;
        };
        // The `cur` words of the staged positions ahead, one per lane, loaded ONE STEP AHEAD: the word of a read's record is the only
        // thing of a move that comes from HBM at the time it is evaluated (the staging wavefront touched the record some tens of
        // draws earlier; next to the greedy chains of the following locus the line has left the L2 by then), and it was a round
        // trip of its own in front of the table gathers. pf of lane l is the word of position pf_base + l; the step that applied a
        // move after the words were requested patches the one it changed (lm_*).
        uint32_t pf = 0, pf_slot = 0xFFFFFFFFu, pf_base = 0, pf_n = 0, lm_slot = 0xFFFFFFFFu, lm_to = 0;
        auto pf_issue = [&]() {                                               // one load per lane, no branch around it
            const uint32_t avail = __hip_atomic_load(&ring->produced, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - consumed;
            pf_base = consumed; pf_n = min(avail, 64u);
            pf_slot = lane < pf_n ? slot_at(lane) : 0u;
            pf = load_rp_cur(&recs[pf_slot]);
            lm_slot = 0xFFFFFFFFu;
        };
        auto pf_covered = [&]() -> uint32_t {                                 // positions from `consumed` on whose words are here
            const uint32_t end = pf_base + pf_n;
            return end - consumed <= 64u ? end - consumed : 0u;                // (unsigned: end < consumed wraps to a large number)
        };
        auto pf_take = [&](uint32_t off) -> uint32_t {                        // every lane calls it; valid for off < pf_covered()
            const int src = static_cast<int>((consumed + off - pf_base) & 63u);
            const uint32_t v = static_cast<uint32_t>(__shfl(static_cast<int>(pf), src));
            const uint32_t vs = static_cast<uint32_t>(__shfl(static_cast<int>(pf_slot), src));
            return vs == lm_slot ? (v & 0xFFFFFFu) | (lm_to << 24) : v;
        };
        auto retire = [&](uint32_t q) {                                    // the producer may reuse the ring entries of q draws
            consumed += q;
            __hip_atomic_store(&ring->consumed, consumed, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        auto blank = [](Move& m) {
            m.slot = 0; m.rp = 0; m.new_assgn = 0; m.ddiff = 0.0;
            m.w1 = m.w2 = m.w3 = m.w4 = 0; m.lp_old = m.lp_new = 0.0;
        };

        // max_abs_random (stoch.rs:19-22) over INIT_ITER = 100 random targets
        double max_abs = 0.0;
        for (uint32_t i = 0; i < 100 && !lost; i++) {
            if (!ring_wait(2)) break;
            Move m; blank(m);
            const uint32_t c = ring_move(0, m);
            max_abs = fmax(max_abs, fabs(improvement(m)));
            retire(c);
        }
        const double min_diff = fmax(1e-10 * max_abs, 1e-14);             // minimum_allowed_diff (stoch.rs:27-29)
        const double start_temp = fmax(-max_abs / log(V.solver.init_prob), 1e-5);
        const double temp_step = start_temp / static_cast<double>(V.solver.anneal_steps);
        uint32_t curr_plato = 0;
        for (uint32_t i = V.solver.anneal_steps; i >= 1 && !lost; i--) {
            if (!ring_wait(3)) break;
            n_iter++;
            Move m; blank(m);
            if (pf_covered() < 1) pf_issue();                                  // the first step (or a ring that ran dry): wait for the words
            uint32_t c = move_pre(0, pf_take(0), m);
            {
                typename ChainT::DepthGather g;
                C.request(m.w1, m.w2, m.w3, m.w4, g);
                pf_issue();                                                     // the words of the next step travel with this step's gathers
                m.ddiff = C.finish(g);
            }
            const double diff = improvement(m) - min_diff;
            bool accept = diff >= 0.0;
            if (!accept) { accept = ring_f64(c) <= exp(diff / (temp_step * static_cast<double>(i))); c++; }
            if (accept) { reassign(m); curr_plato = 0; lm_slot = m.slot; lm_to = m.new_assgn; }
            else { curr_plato++; }
            retire(c);
            if (!accept && curr_plato >= V.solver.plato_size) break;
        }
        // Second loop of stoch.rs:228-241: a move changes the state only when it is accepted, so the moves that
        // follow a rejection see the same state. Lane q evaluates the move that starts at draw q of the random
        // stream (a move takes one draw, two when the read has more than two locations); the lanes that lie on
        // the true chain of moves are then walked in order up to the first accepted one, which is applied, and
        // the stream continues right behind it. Same moves, same order, same result as the serial loop.
        uint64_t iter = 0;
        uint32_t width = 16;                                               // lanes that speculate: about twice the recent run length
        while (!lost && iter < max_iter && curr_plato < V.solver.plato_size) {
            const uint32_t avail = ring_wait(2);                           // a move may take the draw after its own
            if (!avail) break;
            if (pf_covered() < 2) pf_issue();
            const uint32_t w = min(min(width, avail - 1), pf_covered());       // only positions whose `cur` words are here
            Move m; blank(m);
            bool accepted = false, wide = false;
            const uint32_t packed = pf_take(lane);
            if (lane < w) wide = move_pre(lane, packed, m) == 2;                // lanes beyond: windows 0, no effect
            {
                typename ChainT::DepthGather g;
                C.request(m.w1, m.w2, m.w3, m.w4, g);
                pf_issue();                                                     // the words of the next round travel with this round's gathers
                m.ddiff = C.finish(g);
            }
            accepted = lane < w && improvement(m) > min_diff;
            const unsigned long long acc = __ballot(accepted);
            const unsigned long long two = __ballot(wide);
            uint32_t q = 0, walked = 0;
            int hit = -1;
            // Which lanes lie on the true chain of moves: position 0 does; position p > 0 does unless p - 1 does and its move takes two
            // draws. Behind the nearest position r < p whose move takes one draw (or the start) the chain alternates, so p is on it
            // iff p - (r + 1) is even: one count-leading-zeros per lane instead of a serial walk over the draws.
            const unsigned long long below = lane ? ((1ull << lane) - 1ull) : 0ull;
            const unsigned long long ones = ~two & below;
            const uint32_t after = ones ? 64u - static_cast<uint32_t>(__clzll(static_cast<long long>(ones))) : 0u;      // r + 1
            const unsigned long long chain_mask = __ballot(lane < w && ((lane - after) & 1u) == 0u);
            const unsigned long long hits = acc & chain_mask;
            const uint32_t stop = hits ? static_cast<uint32_t>(__ffsll(static_cast<long long>(hits))) - 1u : w;   // first accepted move on the chain
            const uint32_t moves = static_cast<uint32_t>(__popcll(chain_mask & (stop >= 63u ? ~0ull : ((2ull << stop) - 1ull))));   // evaluated in order, the hit included
            const uint32_t rejected = moves - (hits ? 1u : 0u);
            if (iter + moves <= max_iter && curr_plato + rejected < V.solver.plato_size) {
                // neither cap is reached inside this step: the serial loop would have gone exactly this far
                iter += moves; n_iter += moves; walked = moves; curr_plato += rejected;
                if (hits) { hit = static_cast<int>(stop); q = stop + 1u + static_cast<uint32_t>((two >> stop) & 1ull); }
                else {
                    // everything evaluated was rejected: the draws used are those of the chain's moves among the first w positions
                    const uint32_t last = 63u - static_cast<uint32_t>(__clzll(static_cast<long long>(chain_mask)));
                    q = last + 1u + static_cast<uint32_t>((two >> last) & 1ull);
                }
            } else {
                while (q < w) {                                               // the last steps of a chain: one move at a time
                    if (iter >= max_iter || curr_plato >= V.solver.plato_size) break;
                    iter++; n_iter++; walked++;
                    const uint32_t cons = 1 + static_cast<uint32_t>((two >> q) & 1ull);
                    if ((acc >> q) & 1ull) { hit = static_cast<int>(q); q += cons; break; }
                    curr_plato++;
                    q += cons;
                }
            }
            if (hit >= 0) {
                Move a;
                a.rp = __shfl(m.rp, hit); a.new_assgn = __shfl(m.new_assgn, hit); a.slot = __shfl(m.slot, hit);
                a.w1 = __shfl(m.w1, hit); a.w2 = __shfl(m.w2, hit); a.w3 = __shfl(m.w3, hit); a.w4 = __shfl(m.w4, hit);
                a.lp_old = __shfl(m.lp_old, hit); a.lp_new = __shfl(m.lp_new, hit); a.ddiff = __shfl(m.ddiff, hit);
                reassign(a);
                curr_plato = 0;
                lm_slot = a.slot; lm_to = a.new_assgn;                          // the words requested above were read before this store
            }
            width = min(RING - 1, max(8u, hit >= 0 ? (width + 2 * walked + 4) / 2 : 2 * width));
            retire(q);
        }
    }
    if (!handed_over) __syncthreads();                                   // the producer waits for exactly one hand-over
    __hip_atomic_store(&ring->stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (lane == 0) {
        const double lik = V.depth_contrib * depth_lik + V.aln_contrib * aln_lik;       // assgn.rs:235-237
        V.liks[chain] = (V.priors ? V.priors[gi] : 0.0) + lik;                          // solve.rs:827
        V.parts[4 * chain] = aln_lik; V.parts[4 * chain + 1] = depth_lik; V.parts[4 * chain + 2] = static_cast<double>(n_iter);
        V.parts[4 * chain + 3] = static_cast<double>(n_acc);
    }
}

// count_unexplained_reads (solve.rs:718-729): best_at_contig (locs.rs:605-611) is the likelihood-matrix entry
// the exact solver's assignment into a chain's records: only the word a move changes (StageRunner::solve_exact_batch)
__global__ __launch_bounds__(256) void store_cur_kernel(ChainRec* __restrict__ recs, const uint32_t* __restrict__ words, uint64_t n) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) recs[i].rp_cur = words[i];
}

__global__ __launch_bounds__(256) void count_unexplained_kernel(const uint8_t* __restrict__ status, const double* __restrict__ unmapped,
                                                                const double* __restrict__ matrix, uint64_t n_pairs, uint32_t A,
                                                                const uint16_t* __restrict__ ids, uint32_t ploidy,
                                                                unsigned long long* __restrict__ out) {
    uint32_t mine = 0;
    for (uint64_t r = blockIdx.x * 256ull + threadIdx.x; r < n_pairs; r += static_cast<uint64_t>(gridDim.x) * 256ull) {
        if (status[r] != LCTY_READ_GOOD) continue;
        double best = -INFINITY;
        for (uint32_t i = 0; i < ploidy; i++) best = fmax(best, matrix[r * A + ids[i]]);
        mine += best < unmapped[r] + 1e-8;
    }
    const unsigned long long bal = __ballot(mine != 0);
    if (bal) {
        for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(static_cast<int>(mine), o);
        if ((threadIdx.x & 63u) == 0) atomicAdd(out, static_cast<unsigned long long>(mine));
    }
}

// ---- per-read outputs (GenotypeAlignments::create_counts / ReadAssignment::update_counts, assgn.rs:94-96, 374-378) ----
// number of possible locations of every good read pair on the genotype of chain 0 (read_ixs, assgn.rs:52-60)
template <uint32_t P>
__global__ __launch_bounds__(256) void read_nw_kernel(const SolveView V, uint32_t* __restrict__ nw) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= V.n_good) return;
    Geno<P> G; G.init(V, 0);
    Locs<P> L; locs_init(L, V, g, G);
    nw[g] = L.nw;
}

// counts[read_off[read] + location] += 1 for every attempt. The chains of one genotype share the list of non-trivial
// reads (same reads, same order), only the location byte differs; a read with one location is always at location 0.
__global__ __launch_bounds__(256) void assignment_counts_kernel(const SolveView V, const uint32_t* __restrict__ nw,
                                                                const uint64_t* __restrict__ read_off, uint16_t* __restrict__ counts) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < V.n_good && nw[i] == 1) counts[read_off[i]] = static_cast<uint16_t>(V.attempts);
    if (i < V.c_nnt[0]) {
        // the chains of one genotype have the same non-trivial reads: the parts of chain 0's list are everybody's
        const RecList first{V.recs, V.c_seg, V.seg_reads};
        const uint32_t at = first.place(i);
        const uint32_t rp = V.recs[at].rp_cur & 0xFFFFFFu;
        for (uint32_t a = 0; a < V.attempts; a++) {
            const uint32_t loc = V.recs[static_cast<uint64_t>(a) * V.rstride + at].rp_cur >> 24;
            counts[read_off[rp] + loc] += 1;
        }
    }
}

}  // namespace lcty

using namespace lcty;

// ---------------------------------------------------------------- host: K15 and the stage driver
namespace {

double students_t_cdf(double freedom, double x) {         // statrs StudentsT::cdf (location 0, scale 1)
    if (std::isinf(freedom)) return 0.5 * std::erfc(-x / std::sqrt(2.0));
    const double h = freedom / (freedom + x * x);
    const double ib = 0.5 * math::beta_reg(freedom / 2.0, 0.5, h);
    return x <= 0.0 ? ib : 1.0 - ib;
}

// compare_two_likelihoods (src/solvers/solve.rs:319-336) with the Welch tests of src/math/mod.rs:180-220
double compare_two(double mean1, double var1, uint32_t att1, double mean2, double var2, uint32_t att2) {
    const double simple_norm = mean1 - math::ln_add(mean1, mean2);
    if (std::isnormal(var1) && std::isnormal(var2)) {
        double t_pval;
        if (att1 == att2) {
            const double n = att1, var_sum = var1 + var2;
            const double t_stat = (mean1 - mean2) * std::sqrt(n / var_sum);
            const double freedom = (n - 1.0) * var_sum * var_sum / (var1 * var1 + var2 * var2);
            t_pval = students_t_cdf(freedom, t_stat);
        } else {
            const double n1 = att1, n2 = att2, nv1 = var1 / n1, nv2 = var2 / n2, sum = nv1 + nv2;
            const double t_stat = (mean1 - mean2) / std::sqrt(sum);
            const double freedom = sum * sum / (nv1 * nv1 / (n1 - 1.0) + nv2 * nv2 / (n2 - 1.0));
            t_pval = students_t_cdf(freedom, t_stat);
        }
        return std::fmax(simple_norm, std::log(t_pval));
    }
    return simple_norm;
}

void sort_by_mean(const double* lik_mean, uint64_t* ixs, uint64_t n) {      // sort_indices (solve.rs:418-423)
    std::sort(ixs, ixs + n, [&](uint64_t a, uint64_t b) {
        if (lik_mean[a] != lik_mean[b]) return lik_mean[a] > lik_mean[b];
        return a < b;
    });
}

double ln_sum(const double* v, size_t n) {                // Ln::map_sum (math/mod.rs:62-76)
    if (n == 0) return -std::numeric_limits<double>::infinity();
    if (n == 1) return v[0];
    double m = -std::numeric_limits<double>::infinity();
    for (size_t i = 0; i < n; i++) m = std::fmax(m, v[i]);
    if (std::isinf(m)) return m;
    double s = 0.0;
    for (size_t i = 0; i < n; i++) s += std::exp(v[i] - m);
    return m + std::log(s);
}

}  // namespace

namespace {

// location table + compact "unmapped" column of a scored batch (rebuilt after every lcty_score_reads)
void ensure_solver_tables(lcty_reads* reads) {
    reads->ensure_good_index();
    if (reads->loc_table_valid) return;
    lcty_ctx* ctx = reads->ctx;
    lcty_locus* loc = reads->locus;
    const uint64_t n_good = reads->n_good_cached, A = loc->n_alleles;
    const uint64_t ngp = std::max<uint64_t>(64, (n_good + 63) / 64 * 64);
    const size_t need = static_cast<size_t>(A) * ngp;
    if (n_good >= (1ull << 24)) fail(LCTY_ERR_UNSUPPORTED, "the device solver handles up to 2^24 good read pairs per locus");
    if (reads->d_loc_table.n < need * sizeof(LocCell)) reads->d_loc_table.alloc(need * sizeof(LocCell));
    if (reads->d_loc_ext.n < need) reads->d_loc_ext.alloc(need);
    if (reads->d_loc_unm.n < ngp) reads->d_loc_unm.alloc(ngp);
    reads->ngp = ngp;
    if (n_good) {
        const dim3 grid(static_cast<uint32_t>(ngp / 32), static_cast<uint32_t>((A + 31) / 32));
        ctx->timed(LCTY_K_SOLVE_TABLE, [&] {
            hipLaunchKernelGGL(build_loc_table_kernel, grid, dim3(256), 0, ctx->stream, reads->d_good_ix.p, static_cast<uint32_t>(n_good), ngp,
                               static_cast<uint32_t>(A), reads->d_pa_idx.p, reads->d_pa_off.p, reads->d_pa.p, reads->d_unmapped.p,
                               reinterpret_cast<LocCell*>(reads->d_loc_table.p), reads->d_loc_ext.p, reads->d_loc_unm.p, reads->d_err.p);
        });
        LCTY_HIP(hipGetLastError());
        uint32_t flag = 0;
        reads->d_err.download(&flag, 1, ctx->stream);
        LCTY_HIP(hipStreamSynchronize(ctx->stream));
        if (flag == LCTY_ERR_UNSUPPORTED) {
            uint32_t zero = 0;
            reads->d_err.upload(&zero, 1, ctx->stream);
            fail(LCTY_ERR_UNSUPPORTED, "the device solver handles alleles below 16 Mb and pair-alignment arenas below 2^32 entries");
        }
    }
    reads->loc_table_valid = true;
}

// depth table of the locus, wide enough for `n_good` reads piling up on the shortest contig
void ensure_depth_table(lcty_locus* loc, uint64_t want) {
    uint32_t depth = LCTY_DEPTH_CACHE;
    while (depth < want) depth *= 2;
    if (loc->lut_ext_depth >= depth) return;
    lcty_ctx* ctx = loc->ctx;
    loc->d_lut_ext.alloc(static_cast<size_t>(LCTY_GC_BINS) * depth);
    const uint32_t n = LCTY_GC_BINS * depth;
    hipLaunchKernelGGL(build_depth_table_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, loc->d_depth_lut.p, loc->d_depth_nb.p,
                       static_cast<uint32_t>(loc->prm.n_alt_cn), depth, loc->d_lut_ext.p);
    LCTY_HIP(hipGetLastError());
    loc->lut_ext_depth = depth;
}

// a few microseconds of nothing (one wavefront): lets the workgroups of a kernel launched just before on another stream get resident
__global__ void pause_kernel(uint32_t rounds) {
    for (uint32_t i = 0; i < rounds; i++) __builtin_amdgcn_s_sleep(127);
}

template <uint32_t P>
void launch_init(lcty_ctx* ctx, const SolveView& V, uint32_t nch, size_t lds_init, hipStream_t s) {
    if (lds_init > 48 * 1024)
        LCTY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(solve_init_kernel<P>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(lds_init)));
    ctx->timed(LCTY_K_SOLVE_INIT, [&] { hipLaunchKernelGGL(solve_init_kernel<P>, dim3(nch), dim3(256), lds_init, s, V); }, s);
    LCTY_HIP(hipGetLastError());
}

// LDS of a greedy workgroup: the rows' windows (greedy_lds_windows) and four words per row for the parts of its record list
inline size_t greedy_lds(uint32_t lpc, const SolveView& V, bool lw) {
    return greedy_lds_windows(lpc, V.wstride, V.n_wk, V.n_wc, lw) + static_cast<size_t>(64 / lpc) * (lw ? 2 : 1) * GREEDY_ROW_TAIL * 4;
}
template <uint32_t LPC, bool LW, uint32_t FORM = 0>
void launch_greedy_form(lcty_ctx* ctx, const SolveView& V, uint32_t nch, hipStream_t s) {
    constexpr uint32_t ROWS = (64 / LPC) * (LW ? 2 : 1);
    const size_t lds = greedy_lds(LPC, V, LW);
    if (lds > 48 * 1024)
        LCTY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(greedy_loop_kernel<LPC, LW, FORM>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(lds)));
    ctx->timed(LCTY_K_SOLVE, [&] {
        hipLaunchKernelGGL((greedy_loop_kernel<LPC, LW, FORM>), dim3((nch + ROWS - 1) / ROWS), dim3(LW ? 128 : 64), lds, s, V, nch);
    }, s);
    LCTY_HIP(hipGetLastError());
}
template <uint32_t LPC, bool LW>
void launch_greedy(lcty_ctx* ctx, const SolveView& V, uint32_t nch, hipStream_t s) {
    if constexpr (LPC == 12 && LW) {
        // the forms under measurement exist for the stage shape of the default scheme only
        switch (ctx->knob("solve_greedy_form", 0)) {
            case 32: return launch_greedy_form<LPC, LW, 32>(ctx, V, nch, s);
            default: break;
        }
    }
    launch_greedy_form<LPC, LW, 0>(ctx, V, nch, s);
}

template <int MODE>
void launch_anneal_as(lcty_ctx* ctx, const SolveView& V, uint32_t nch, hipStream_t s) {
    const size_t words = (static_cast<size_t>(V.wstride) * 4 + 31) & ~static_cast<size_t>(31);
    const size_t lds = (MODE == 1 ? static_cast<size_t>(V.wstride) * 8 : MODE == 2 ? static_cast<size_t>(V.n_wk + V.n_wc) * 8 + ((static_cast<size_t>(V.wstride) * 2 + 31) & ~static_cast<size_t>(31)) : 0) +
                       words + sizeof(AnnealRing) + 64;
    if (lds > 48 * 1024)
        LCTY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(anneal_loop_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(lds)));
    ctx->timed(LCTY_K_ANNEAL, [&] { hipLaunchKernelGGL(anneal_loop_kernel<MODE>, dim3(nch), dim3(128), lds, s, V); }, s);
    LCTY_HIP(hipGetLastError());
}
void launch_anneal(lcty_ctx* ctx, const SolveView& V, uint32_t nch, hipStream_t s) {
    // lcty_ctx_set_knob "anneal_lds_weights": 0 gathered, 1 in LDS as they are, 2 table indices + tables in LDS (the kernel's comment);
    // default 2 where the locus has the tables and the depths fit their field, else 0
    const bool tables = V.n_wk != 0 && 2 * static_cast<uint64_t>(V.n_good) + 2 <= LW_DEPTH_MASK;
    int64_t mode = ctx->knob("anneal_lds_weights", tables ? 2 : 0);
    if (mode == 2 && !tables) mode = 0;
    if (mode == 2) launch_anneal_as<2>(ctx, V, nch, s);
    else if (mode == 1) launch_anneal_as<1>(ctx, V, nch, s);
    else launch_anneal_as<0>(ctx, V, nch, s);
}

// One stage = every (genotype, attempt) chain, in batches that fit the state budget. `after_batch(g0, ng, liks)` runs
// while the batch's device state (records of the non-trivial reads, window arrays) is still alive.
struct StageRunner {
    lcty_reads* reads; lcty_ctx* ctx; lcty_locus* loc;
    SolveView V{};
    uint64_t n_gt; uint32_t ploidy, attempts;
    size_t lds_init = 0;
    uint64_t gt_per_batch = 1, depth_cap = 2;
    uint32_t lane;                      // 0: the context's stream; 1: its side stream (the last stage of a locus while the next locus starts)
    hipStream_t stream;
    lcty_ctx::SolveWorkspace& ws;       // device state of the chains: grow-only, lives as long as the context

    StageRunner(lcty_reads* r, const uint16_t* genotypes, uint64_t n_gt_, uint32_t ploidy_, const lcty_solver* solver, uint32_t attempts_,
                const uint64_t* chain_seeds, uint32_t lane_ = 0, const RowGatherer* gathered = nullptr)
        : reads(r), n_gt(n_gt_), ploidy(ploidy_), attempts(attempts_), lane(lane_), ws(check_args(r, genotypes, solver, chain_seeds, lane_)) {
        if (!reads->scored) fail(LCTY_ERR_INVALID_INPUT, "lcty_score_reads has not been called on this batch");
        if (ploidy == 0 || ploidy > MAXP) fail(LCTY_ERR_UNSUPPORTED, "the device solver handles ploidy 1..%u", MAXP);
        if (attempts == 0) fail(LCTY_ERR_INVALID_INPUT, "At least one attempt is required for each stage");
        if (solver->kind != LCTY_SOLVER_GREEDY && solver->kind != LCTY_SOLVER_ANNEAL && solver->kind != LCTY_SOLVER_EXACT)
            fail(LCTY_ERR_INVALID_INPUT, "unknown solver kind");
        if (solver->kind == LCTY_SOLVER_ANNEAL && !(solver->init_prob > 0.0 && solver->init_prob <= 1.0))
            fail(LCTY_ERR_INVALID_INPUT, "Initial probability (%g) must be within (0, 1]", solver->init_prob);
        if (solver->kind == LCTY_SOLVER_ANNEAL && solver->anneal_steps == 0) fail(LCTY_ERR_INVALID_INPUT, "Number of annealing steps must be positive");
        if (solver->kind == LCTY_SOLVER_GREEDY && solver->sample_size == 0) fail(LCTY_ERR_INVALID_INPUT, "Sample size must be positive");
        if (solver->kind == LCTY_SOLVER_GREEDY && solver->sample_size > 64) fail(LCTY_ERR_UNSUPPORTED, "greedy sample size above 64");
        ctx = reads->ctx; loc = reads->locus;
        ctx->activate();
        stream = lane ? ctx->side_stream() : ctx->stream;
        reads->check_device_error();
        const uint32_t A = loc->n_alleles;
        for (uint64_t i = 0; i < n_gt * ploidy; i++)
            if (genotypes[i] >= A) fail(LCTY_ERR_INVALID_INPUT, "genotype refers to allele %u >= %u", genotypes[i], A);
        if (n_gt * attempts >= 0x7FFFFFFFull) fail(LCTY_ERR_UNSUPPORTED, "too many chains in one stage");
        if (!gathered) ensure_solver_tables(reads);
        reads->stat_chains = reads->stat_iterations = reads->stat_accepted = 0;
        // the batch's own table, or the rows of the stage's alleles over the reads of every shard of the locus
        const uint64_t n_good = gathered ? gathered->n_good : reads->n_good_cached, ngp = gathered ? gathered->ngp : reads->ngp;
        // depth table: wide enough for twice the mean depth of "every read on the shortest contig" (two mates per pair);
        // a chain that still runs past it raises `overflow` and the batch is repeated with a wider table
        uint32_t min_w = 0xFFFFFFFFu;
        for (uint32_t a = 0; a < A; a++) min_w = std::min(min_w, std::max(loc->n_windows[a], 1u));
        depth_cap = 2 * n_good + 2;                                      // no window can be deeper
        uint64_t first_width = std::min<uint64_t>(4 * n_good / min_w + 64, depth_cap);
        if (ctx->knob("depth_table_start", 0) > 0)                          // lcty_ctx_set_knob: start narrow, exercise the widening
            first_width = static_cast<uint64_t>(ctx->knob("depth_table_start", 0));
        ensure_depth_table(loc, first_width);

        V.by_window = FastDiv::make(loc->bg.window); V.by_tweak = FastDiv::make(2 * static_cast<uint32_t>(loc->prm.tweak) + 1);
        V.A = A; V.window = loc->bg.window; V.left_padding = loc->left_padding; V.tweak = static_cast<uint32_t>(loc->prm.tweak);
        V.min_weight = loc->prm.min_weight; V.prob_diff = loc->prm.prob_diff;
        V.depth_contrib = 1.0 + loc->prm.lik_skew; V.aln_contrib = 1.0 - loc->prm.lik_skew;      // assgn.rs:80-81
        V.n_windows = loc->d_n_windows.p; V.reg_start = loc->d_reg_start.p; V.allele_len = loc->d_allele_len.p;
        V.ci_off = loc->d_ci_off.p; V.gc = loc->d_gc.p; V.win_weight = loc->d_win_weight.p;
        V.uniq_cnt = loc->d_uniq_cnt.p; V.compl_cnt = loc->d_compl_cnt.p;
        const bool tables = loc->weight_tables_valid && !loc->has_explicit;
        V.wk = tables ? loc->d_wk.p : nullptr; V.wc = tables ? loc->d_wc.p : nullptr;
        V.n_wk = tables ? static_cast<uint32_t>(loc->d_wk.n) : 0u; V.n_wc = tables ? static_cast<uint32_t>(loc->d_wc.n) : 0u;
        V.lut = loc->d_lut_ext.p; V.lut_depth = loc->lut_ext_depth; V.lut_shift = static_cast<uint32_t>(__builtin_ctz(loc->lut_ext_depth)); V.depth_nb = loc->d_depth_nb.p; V.n_alt = loc->prm.n_alt_cn;
        V.n_good = static_cast<uint32_t>(n_good); V.ngp = ngp;
        V.seg_reads = static_cast<uint32_t>(((n_good + INIT_SEGS - 1) / INIT_SEGS + 63) / 64 * 64);      // the parts of a chain's record list
        if (V.seg_reads == 0) V.seg_reads = 64;
        V.rstride = static_cast<uint64_t>(INIT_SEGS) * V.seg_reads;
        V.table = reinterpret_cast<const LocCell*>(reads->d_loc_table.p); V.table_ext = reads->d_loc_ext.p; V.table_unm = reads->d_loc_unm.p;
        V.pa = reads->d_pa.p; V.row_of = nullptr;
        if (gathered) {
            V.table = reinterpret_cast<const LocCell*>(reads->gather.table.p); V.table_ext = reads->gather.ext.p; V.table_unm = reads->gather.unm.p;
            V.pa = reads->gather.pa.p; V.row_of = reads->gather.row_of.p;
        }
        V.ploidy = ploidy; V.attempts = attempts; V.solver = *solver;
        V.wstride = (2 + ploidy * loc->max_n_windows + 3) & ~3u;
        V.prio_mode = static_cast<uint32_t>(ctx->knob("solve_prio_mode", 0));
        lds_init = ((static_cast<size_t>(V.wstride) * 4 + 15) & ~static_cast<size_t>(15)) + 256 * 8 + 64;
        if (static_cast<size_t>(V.wstride) * 12 + sizeof(AnnealRing) + 128 > 160 * 1024 || V.wstride > 65535)
            fail(LCTY_ERR_UNSUPPORTED, "%u windows per genotype: too many for the device solver", V.wstride);

        // Locations beyond the second of a read (ploidy > 2, several pair-alignments on a contig, "both unmapped" in reach): a run per
        // chain; a chain that needs more raises a flag and the batch is repeated with the run it asked for
        // (the run size a stage asked for is kept for the next stages and loci of the context: loci of one data set look alike)
        {
            const uint32_t guess = static_cast<uint32_t>(std::min<uint64_t>(ngp * (ploidy > 2 ? ploidy - 2 : 0) + std::max<uint64_t>(256, ngp / 64), (1u << 24) - 1));
            if (ws.extra_for_ploidy != ploidy) ws.extra_cap = 0;
            ws.extra_for_ploidy = ploidy;
            if (ctx->knob("solve_extra_start", 0) > 0) {                          // tests: exercise the growth
                if (ws.extra_cap == 0) ws.extra_cap = static_cast<uint32_t>(ctx->knob("solve_extra_start", 0));
            } else ws.extra_cap = std::max(ws.extra_cap, guess);
        }
        plan_batches();
        V.overflow = ws.ovf.p;
    }

    static lcty_ctx::SolveWorkspace& check_args(lcty_reads* r, const uint16_t* genotypes, const lcty_solver* solver, const uint64_t* chain_seeds,
                                                uint32_t lane) {
        if (!r || !genotypes || !solver || !chain_seeds) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        return r->ctx->solve_ws[lane ? 1 : 0];
    }

    // chains are processed in batches so that the per-chain state (32 B per good read + the run of further locations) fits the device
    void plan_batches() {
        // one lane at a time: a lane that read the free memory while the other had just released its workspace to enlarge it would
        // count that memory as its own
        std::lock_guard<std::mutex> ws_lock(ctx->ws_mutex);
        const uint64_t ngp = V.rstride;                                     // record places per chain
        const uint64_t per_chain = (ngp + ngp / 128) * sizeof(ChainRec) + static_cast<uint64_t>(ws.extra_cap) * sizeof(ExtraLoc) + static_cast<uint64_t>(V.wstride) * 21 + 64;
        size_t free_b = 0, total_b = 0;
        ctx->release_transfer_scratch();
        LCTY_HIP(hipMemGetInfo(&free_b, &total_b));
        const uint64_t held = ws.recs.n * sizeof(ChainRec) + ws.extra.n * sizeof(ExtraLoc);      // what this workspace already owns counts as free
        uint64_t budget = static_cast<uint64_t>(0.92 * static_cast<double>(free_b + held));
        if (ctx->knob("solve_budget_mb", 0) > 0)                              // lcty_ctx_set_knob: force several batches
            budget = static_cast<uint64_t>(ctx->knob("solve_budget_mb", 0)) << 20;
        gt_per_batch = std::max<uint64_t>(1, std::min<uint64_t>(n_gt, budget / (per_chain * attempts)));
        const uint64_t max_chains = gt_per_batch * attempts;
        hipStream_t s = stream;
        ws.ovf.ensure(2); ws.ovf.zero(s);
        if (ws.recs.n < max_chains * ngp || ws.extra.n < max_chains * ws.extra_cap + 2) {
            // both at once, the old ones released first: the two together are most of the device
            if (ctx->knob("queue_trace", 0))
                fprintf(stderr, "[lcty queue] lane %u workspace: %llu chains x %llu places (had %.1f GB of records, %.1f GB of runs; free %.1f GB, budget %.1f GB, %u further locations per chain)\n",
                        lane, static_cast<unsigned long long>(max_chains), static_cast<unsigned long long>(ngp), ws.recs.n * 32e-9, ws.extra.n * 16e-9,
                        free_b * 1e-9, budget * 1e-9, ws.extra_cap);
            // Grow-only, with a little head-room: the loci of a queue differ by a fraction of a per cent in their good read pairs, and a
            // workspace that followed every locus exactly was released and allocated again (4 s for 150 GB, with every stream of the
            // device waiting) whenever a slightly larger locus came after a smaller one.
            const uint64_t want_recs = std::max<uint64_t>(ws.recs.n, max_chains * (ngp + ngp / 128));
            const uint64_t want_extra = std::max<uint64_t>(ws.extra.n, max_chains * static_cast<uint64_t>(ws.extra_cap) + 2);   // two spare entries: the greedy loop reads a pair per record
            ws.recs.release(); ws.extra.release();
            ws.recs.alloc(want_recs); ws.extra.alloc(want_extra);
        }
        ws.cww.ensure(max_chains * V.wstride); ws.cgc.ensure(max_chains * V.wstride); ws.cdepth.ensure(max_chains * V.wstride);
        ws.cuc.ensure(max_chains * V.wstride);
        ws.cnnt.ensure(max_chains); ws.cseg.ensure(4 * max_chains); ws.ctotw.ensure(max_chains); ws.caln.ensure(max_chains);
        ws.gt.ensure(gt_per_batch * ploidy); ws.seeds.ensure(max_chains); ws.liks.ensure(max_chains); ws.parts.ensure(4 * max_chains);
        ws.pri.ensure(gt_per_batch);
        V.genotypes = ws.gt.p; V.seeds = ws.seeds.p; V.priors = nullptr;
        V.recs = ws.recs.p; V.extra = ws.extra.p; V.extra_cap = ws.extra_cap; V.liks = ws.liks.p; V.parts = ws.parts.p;
        V.c_ww = ws.cww.p; V.c_uc = ws.cuc.p; V.c_gc = ws.cgc.p; V.c_depth = ws.cdepth.p; V.c_nnt = ws.cnnt.p; V.c_seg = ws.cseg.p; V.c_totw = ws.ctotw.p; V.c_aln = ws.caln.p;
    }

    void upload_genotypes(const uint16_t* genotypes, uint64_t ng) { ws.gt.upload(genotypes, ng * ploidy, stream); }

    void launch(uint32_t nch) {
        switch (ploidy) {
            case 1: launch_init<1>(ctx, V, nch, lds_init, stream); break;
            case 2: launch_init<2>(ctx, V, nch, lds_init, stream); break;
            case 3: launch_init<3>(ctx, V, nch, lds_init, stream); break;
            default: launch_init<4>(ctx, V, nch, lds_init, stream); break;
        }
        if (V.solver.kind == LCTY_SOLVER_EXACT) { solve_exact_batch(nch); return; }
        if (V.solver.kind == LCTY_SOLVER_ANNEAL) {
            if (lane == 1 && !ctx->knob("queue_no_gate", 0)) wait_for_greedy_of_next_locus();
            launch_anneal(ctx, V, nch, stream);
            if (ctx->knob("queue_trace", 0)) fprintf(stderr, "[lcty queue] %.3f ms batch %p annealing launched (lane %u)\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(), static_cast<const void*>(reads), lane);
            return;
        }
        if (lane == 0) announce_greedy();
        if (ctx->knob("queue_trace", 0)) fprintf(stderr, "[lcty queue] %.3f ms batch %p greedy about to launch (lane %u)\n",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(), static_cast<const void*>(reads), lane);
        // Lanes per chain. The loop is bound by instruction issue — a wavefront-iteration costs the same whatever its number of busy
        // lanes — and at its register count one wavefront fits a SIMD: the chains of a stage should make at most one wavefront per
        // SIMD. A row of 16 lanes (the hardware's DPP rows: cheapest row operations) holds the default sample of 10 with four
        // chains per wavefront; when that makes more wavefronts than SIMDs, rows of 12 or 10 lanes put five or six chains into one.
        const uint32_t S = V.solver.sample_size;
        const uint64_t simds = 4ull * static_cast<uint64_t>(ctx->props.multiProcessorCount);
        uint32_t lpc = S <= 16 ? 16 : S <= 32 ? 32 : 64;
        if (lpc == 16 && (nch + 3) / 4 > simds) {
            if (S <= 12 && (nch + 4) / 5 <= simds) lpc = 12;
            else if (S <= 10) lpc = 10;
            else if (S <= 12) lpc = 12;
        }
        const int64_t want = ctx->knob("solve_chains_per_wave", 0);
        if (want > 0) {                                                        // lcty_ctx_set_knob: 1, 2, 4, 5, 6 chains per wavefront
            const uint32_t forced = want >= 6 ? 10u : want == 5 ? 12u : static_cast<uint32_t>(64 / std::min<int64_t>(want, 4));
            if (forced >= S) lpc = forced;
        }
        // window weights from LDS tables when the locus has them (no explicit weights, counts within the index bits) and the
        // depths fit their field; lcty_ctx_set_knob "solve_lds_weights" 0 keeps the gathers
        const bool lw = V.n_wk != 0 && 2 * static_cast<uint64_t>(V.n_good) + 2 <= LW_DEPTH_MASK && ctx->knob("solve_lds_weights", 1) != 0;
        while (lpc < 64 && greedy_lds(lpc, V, lw) > 80 * 1024) lpc = lpc < 16 ? 16 : lpc * 2;      // the rows' windows share the LDS
        auto go = [&](auto tag) {
            constexpr uint32_t L = decltype(tag)::value;
            if (lw) launch_greedy<L, true>(ctx, V, nch, stream); else launch_greedy<L, false>(ctx, V, nch, stream);
        };
        const bool timed_form = lpc == 12 && lw && (ctx->knob("solve_greedy_form", 0) & 32) != 0;
        const size_t n_waves = (static_cast<size_t>(nch) + 4) / 5 + 2;
        V.dbg = nullptr;
        if (timed_form) { ws.dbg.ensure(12 * n_waves); ws.dbg.zero(stream); V.dbg = ws.dbg.p; }
        if (lpc == 10) go(std::integral_constant<uint32_t, 10>{});
        else if (lpc == 12) go(std::integral_constant<uint32_t, 12>{});
        else if (lpc == 16) go(std::integral_constant<uint32_t, 16>{});
        else if (lpc == 32) go(std::integral_constant<uint32_t, 32>{});
        else go(std::integral_constant<uint32_t, 64>{});
        if (timed_form) {
            // diagnostic: mean shader-clock cycles per iteration and phase over the wavefronts of the launch
            std::vector<double> d(12 * n_waves);
            ws.dbg.download(d.data(), d.size(), stream);
            LCTY_HIP(hipStreamSynchronize(stream));
            double sum[12] = {0}; size_t used = 0;
            for (size_t w = 0; w < n_waves; w++) if (d[12 * w + 6] > 0) { used++; for (int k = 0; k < 12; k++) sum[k] += d[12 * w + k] / d[12 * w + 6]; }
            if (used) fprintf(stderr, "[lcty greedy phases] %zu wavefronts; clock ticks per iteration: candidates+requests %.0f (waiting for the further locations %.0f, current location %.0f, its pair %.0f, first alternative %.0f), ext+sample+record request %.0f, wait+score %.0f, row best %.0f, move %.0f; loop total %.0f\n",
                              used, sum[0] / used, sum[11] / used, sum[8] / used, sum[9] / used, sum[10] / used, sum[1] / used, sum[2] / used, sum[3] / used, sum[4] / used, sum[5] / used);
        }
    }

    // lcty_ctx::LaunchGate: the main stream's greedy loop of the next locus goes first, the side stream's annealing loop right behind
    void announce_greedy() {
        auto& g = ctx->gate;
        if (!g.ev) LCTY_HIP(hipEventCreateWithFlags(&g.ev, hipEventDisableTiming));
        {
            std::lock_guard<std::mutex> lock(g.m);
            LCTY_HIP(hipEventRecord(g.ev, stream));                              // behind the initialisation kernel: the greedy loop is next
            g.epoch++;
        }
        g.cv.notify_all();
    }
    void wait_for_greedy_of_next_locus() {
        auto& g = ctx->gate;
        std::unique_lock<std::mutex> lock(g.m);
        if (g.target == 0) return;
        g.cv.wait(lock, [&] { return g.epoch >= g.target; });
        if (g.ev) {
            LCTY_HIP(hipStreamWaitEvent(stream, g.ev, 0));
            hipLaunchKernelGGL(pause_kernel, dim3(1), dim3(64), 0, stream, 8u);
        }
        g.target = 0;
    }

    // ---- the exact solver (src/solvers/highs.rs:38-134, gurobi.rs:15-83) ----
    // SOLVED ON THE HOST (lcty_exact.cpp: branch and bound under a Lagrangian bound). The model of a chain is what solve_init_kernel has
    // just built on the device (records = the columns of the reads with their objective and windows after apply_tweak, the window arrays
    // = the depth distributions); the models of a group of chains are brought to the host, solved by a pool of host threads — one model
    // per thread at a time, as the reference runs one model per worker (solve.rs:1052-1062) — and the assignments go back into the
    // chains' records, so per-read counts and BAM output see them like any other solver's. With tweak = 0 apply_tweak draws nothing and
    // the attempts of a genotype share one model: it is solved once. `node_limit` nodes without a proof of optimality (within the
    // relative gap the caller allows, HiGHS' mip_rel_gap) -> LCTY_ERR_SOLVER, as a non-optimal HiGHS status is (highs.rs:113-116).
    void solve_exact_batch(uint32_t nch) {
        hipStream_t s = stream;
        uint32_t ovf[2] = {0, 0};
        ws.ovf.download(ovf, 2, s);
        LCTY_HIP(hipStreamSynchronize(s));
        if (ovf[0]) return;                                                     // run() repeats the batch (wider table / longer runs)
        const uint32_t W = V.wstride;
        const uint32_t ng = (nch + attempts - 1) / attempts;
        std::vector<uint32_t> nnt(nch), totw(nch), seg(4ull * nch);
        std::vector<double> aln0(nch), liks(nch), parts(4ull * nch, 0.0);
        std::vector<uint16_t> gids(static_cast<size_t>(ng) * ploidy);
        ws.cnnt.download(nnt.data(), nch, s); ws.ctotw.download(totw.data(), nch, s); ws.cseg.download(seg.data(), 4ull * nch, s);
        ws.caln.download(aln0.data(), nch, s); ws.gt.download(gids.data(), gids.size(), s);
        std::vector<double> pri(gt_per_batch, 0.0);
        if (V.priors) ws.pri.download(pri.data(), ng, s);
        LCTY_HIP(hipStreamSynchronize(s));
        const bool shared_model = V.tweak == 0 && attempts > 1;                 // one model per genotype
        std::vector<uint32_t> todo;                                              // the chains whose model is solved
        for (uint32_t c = 0; c < nch; c++) if (!shared_model || c % attempts == 0) todo.push_back(c);
        // a group of models at a time: what the host holds of them (records, runs, window arrays) stays below ~2 GB
        const size_t model_bytes = static_cast<size_t>(V.rstride) * sizeof(ChainRec) + static_cast<size_t>(V.extra_cap) * sizeof(ExtraLoc) + static_cast<size_t>(W) * 13 + 4096;
        const size_t group = std::max<size_t>(1, std::min<size_t>(todo.size(), (2ull << 30) / model_bytes));
        const uint32_t hw = std::max(1u, std::thread::hardware_concurrency());
        const uint32_t n_threads = static_cast<uint32_t>(std::max<int64_t>(1, std::min<int64_t>(ctx->knob("exact_threads", std::min(hw, 64u)), 256)));
        const int trace = static_cast<int>(ctx->knob("exact_trace", 0));
        std::vector<double> lut;
        struct Held { std::vector<ChainRec> recs; std::vector<uint32_t> place; exact::Model model; exact::Result res; };
        for (size_t g0 = 0; g0 < todo.size(); g0 += group) {
            const size_t gn = std::min(group, todo.size() - g0);
            std::vector<Held> held(gn);
            std::vector<ExtraLoc> extra(std::max<uint32_t>(V.extra_cap, 1));
            std::vector<uint32_t> depth0(W);
            uint64_t need = 0;
            for (size_t k = 0; k < gn; k++) {
                const uint32_t c = todo[g0 + k], n = nnt[c], tw = totw[c];
                Held& h = held[k];
                exact::Model& m = h.model;
                h.recs.resize(V.rstride); m.ww.resize(W); m.gcb.resize(W);
                LCTY_HIP(hipMemcpyAsync(h.recs.data(), V.recs + static_cast<uint64_t>(c) * V.rstride, V.rstride * sizeof(ChainRec), hipMemcpyDeviceToHost, s));
                LCTY_HIP(hipMemcpyAsync(extra.data(), V.extra + static_cast<uint64_t>(c) * V.extra_cap, static_cast<size_t>(V.extra_cap) * sizeof(ExtraLoc), hipMemcpyDeviceToHost, s));
                LCTY_HIP(hipMemcpyAsync(m.ww.data(), V.c_ww + static_cast<uint64_t>(c) * W, W * sizeof(double), hipMemcpyDeviceToHost, s));
                LCTY_HIP(hipMemcpyAsync(m.gcb.data(), V.c_gc + static_cast<uint64_t>(c) * W, W, hipMemcpyDeviceToHost, s));
                LCTY_HIP(hipMemcpyAsync(depth0.data(), V.c_depth + static_cast<uint64_t>(c) * W, W * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                LCTY_HIP(hipStreamSynchronize(s));
                m.n = n; m.tw = tw; m.aln0 = aln0[c];
                m.ww.resize(tw); m.gcb.resize(tw); m.depth0.assign(depth0.begin(), depth0.begin() + tw);
                m.first.assign(n + 1, 0); h.place.resize(n);
                const uint32_t* cum = &seg[4ull * c];
                for (uint32_t i = 0; i < n; i++) {
                    const uint32_t q = (i >= cum[1]) + (i >= cum[2]) + (i >= cum[3]);
                    h.place[i] = i - cum[q] + q * V.seg_reads;
                    const ChainRec& r = h.recs[h.place[i]];
                    const uint32_t nloc = r.meta & 0xFFu, eix = r.meta >> 8;
                    m.first[i] = static_cast<uint32_t>(m.locs.size());
                    for (uint32_t t = 0; t < nloc; t++) {
                        if (t == 0) m.locs.push_back({r.lp0, r.win0 & 0xFFFFu, r.win0 >> 16});
                        else if (t == 1) m.locs.push_back({r.lp1, r.win1 & 0xFFFFu, r.win1 >> 16});
                        else { const ExtraLoc& e = extra[eix + t - 2]; m.locs.push_back({e.lp, e.win & 0xFFFFu, e.win >> 16}); }
                    }
                }
                m.first[n] = static_cast<uint32_t>(m.locs.size());
                m.allele_first_w.assign(1, 2u);
                for (uint32_t q = 0; q < ploidy; q++)
                    m.allele_first_w.push_back(m.allele_first_w.back() + loc->n_windows[gids[static_cast<size_t>(c / attempts) * ploidy + q]]);
                m.aln_contrib = V.aln_contrib; m.depth_contrib = V.depth_contrib;
                m.node_limit = V.solver.node_limit ? V.solver.node_limit : 20ull * 1000 * 1000;
                m.rel_gap = V.solver.init_prob > 0.0 && V.solver.init_prob < 1.0 ? V.solver.init_prob : 0.0;
                m.chain = c; m.trace = trace; m.gc_bins = LCTY_GC_BINS;
                if (c == 0) m.dump_path = ctx->exact_dump_path;
                need = std::max(need, exact::depth_needed(m));
            }
            if (need > loc->lut_ext_depth) {
                std::lock_guard<std::mutex> ws_lock(ctx->ws_mutex);             // the other lane of a queue may be sizing its own stage
                ensure_depth_table(loc, std::min<uint64_t>(need, depth_cap));
                V.lut = loc->d_lut_ext.p; V.lut_depth = loc->lut_ext_depth; lut.clear();
            }
            if (lut.empty()) {
                lut.resize(static_cast<size_t>(LCTY_GC_BINS) * loc->lut_ext_depth);
                loc->d_lut_ext.download(lut.data(), lut.size(), s);
                LCTY_HIP(hipStreamSynchronize(s));
            }
            const uint32_t ld = loc->lut_ext_depth;
            // the pool: a worker takes the next model of the group (the largest first would balance better; the models of a stage are alike)
            std::atomic<size_t> next{0};
            auto work = [&] { for (size_t k; (k = next.fetch_add(1)) < gn;) exact::solve(held[k].model, lut.data(), ld, held[k].res); };
            const uint32_t nt = static_cast<uint32_t>(std::min<size_t>(n_threads, gn));
            if (nt <= 1) work();
            else {
                std::vector<std::thread> pool;
                for (uint32_t t = 0; t < nt; t++) pool.emplace_back(work);
                for (auto& th : pool) th.join();
            }
            for (size_t k = 0; k < gn; k++) {
                const Held& h = held[k];
                if (h.res.out_of_nodes)
                    fail(LCTY_ERR_SOLVER, "Exact solver: no proof of optimality within %llu nodes (%u non-trivial reads, %u of them free after fixing the dominated ones); Model finished with non-optimal status NodeLimit",
                         static_cast<unsigned long long>(h.model.node_limit), h.model.n, h.res.n_free);
            }
            // the assignments back into the records (of every attempt that shares the model: same reads at the same places, only the
            // runs of further locations are laid out per chain, so only the `cur` words travel); the likelihood as ReadAssignment::likelihood sums it
            DevBuf<uint32_t> d_words;
            d_words.alloc(V.rstride);
            std::vector<uint32_t> words(V.rstride);
            for (size_t k = 0; k < gn; k++) {
                Held& h = held[k];
                const uint32_t c0 = todo[g0 + k];
                for (uint64_t j = 0; j < V.rstride; j++) words[j] = h.recs[j].rp_cur;
                for (uint32_t i = 0; i < h.model.n; i++) words[h.place[i]] = (words[h.place[i]] & 0xFFFFFFu) | (static_cast<uint32_t>(h.res.assign[i]) << 24);
                d_words.upload(words.data(), V.rstride, s);
                const uint32_t c1 = shared_model ? std::min(c0 + attempts, nch) : c0 + 1;
                for (uint32_t c = c0; c < c1; c++) {
                    if (nnt[c] != h.model.n) fail(LCTY_ERR_RUNTIME, "exact solver: the attempts of a genotype without a tweak differ in their models");
                    hipLaunchKernelGGL(store_cur_kernel, dim3(static_cast<uint32_t>((V.rstride + 255) / 256)), dim3(256), 0, s,
                                       V.recs + static_cast<uint64_t>(c) * V.rstride, d_words.p, V.rstride);
                    LCTY_HIP(hipGetLastError());
                    liks[c] = pri[c / attempts] + h.res.value;
                    parts[4ull * c] = h.res.aln_lik; parts[4ull * c + 1] = h.res.depth_lik; parts[4ull * c + 2] = static_cast<double>(h.res.nodes); parts[4ull * c + 3] = 0.0;
                }
                LCTY_HIP(hipStreamSynchronize(s));                              // `words` is filled again for the next model
            }
            LCTY_HIP(hipStreamSynchronize(s));                                  // the records leave `held` with the group
        }
        ws.liks.upload(liks.data(), nch, s);
        ws.parts.upload(parts.data(), 4ull * nch, s);
        LCTY_HIP(hipStreamSynchronize(s));
    }

    template <typename F>
    void run(const uint16_t* genotypes, const double* priors, const uint64_t* chain_seeds, F&& after_batch) {
        hipStream_t s = stream;
        std::vector<double> liks(gt_per_batch * attempts);
        for (uint64_t g0 = 0; g0 < n_gt; g0 += gt_per_batch) {
            const uint64_t ng = std::min(gt_per_batch, n_gt - g0), nch = ng * attempts;
            upload_genotypes(genotypes + g0 * ploidy, ng);
            ws.seeds.upload(chain_seeds + g0 * attempts, nch, s);
            if (priors) ws.pri.upload(priors + g0, ng, s);
            V.priors = priors ? ws.pri.p : nullptr;
            for (;;) {
                V.lut = loc->d_lut_ext.p; V.lut_depth = loc->lut_ext_depth; V.lut_shift = static_cast<uint32_t>(__builtin_ctz(loc->lut_ext_depth));
                launch(static_cast<uint32_t>(nch));
                uint32_t ovf[2] = {0, 0};
                ws.ovf.download(ovf, 2, s);
                ws.liks.download(liks.data(), nch, s);
                LCTY_HIP(hipStreamSynchronize(s));
                if (!ovf[0]) {
                    std::vector<double> parts(4 * nch);
                    ws.parts.download(parts.data(), 4 * nch, s);
                    LCTY_HIP(hipStreamSynchronize(s));
                    double sum = 0, mx = 0, mn = 1e300, acc = 0;
                    for (uint64_t c = 0; c < nch; c++) {
                        const double it = parts[4 * c + 2];
                        sum += it; mx = std::max(mx, it); mn = std::min(mn, it); acc += parts[4 * c + 3];
                    }
                    reads->stat_chains += nch; reads->stat_iterations += static_cast<uint64_t>(sum);
                    reads->stat_accepted += static_cast<uint64_t>(acc);
                    if (ctx->knob("solve_stats", 0))
                        fprintf(stderr, "[lcty solve] chains=%llu iterations mean=%.0f min=%.0f max=%.0f accepted mean=%.0f lut_depth=%u\n",
                                static_cast<unsigned long long>(nch), sum / nch, mn, mx, acc / nch, loc->lut_ext_depth);
                    break;
                }
                if (ovf[0] == 2) fail(LCTY_ERR_UNSUPPORTED, "a read pair with more than 255 possible locations on one genotype (or 2^24 further locations in a chain)");
                if (ovf[0] == 3) fail(LCTY_ERR_RUNTIME, "annealing kernel: the staging wavefront and the chain lost each other");
                ws.ovf.zero(s);
                if (ovf[0] == 4) {
                    // a chain has more locations beyond the second than its run holds: the batch again with the run it asked for
                    if (ovf[1] >= (1u << 24)) fail(LCTY_ERR_UNSUPPORTED, "2^24 or more further locations in one chain");
                    ws.extra_cap = std::min<uint32_t>(std::max<uint32_t>(ovf[1] + ovf[1] / 8 + 64, 2 * ws.extra_cap), (1u << 24) - 1);
                    const uint64_t before = gt_per_batch;
                    plan_batches();
                    V.overflow = ws.ovf.p;
                    V.priors = priors ? ws.pri.p : nullptr;                    // plan_batches starts from "no priors"; this batch's are uploaded
                    if (gt_per_batch < ng) fail(LCTY_ERR_RUNTIME, "device memory: %llu chains of this stage do not fit with %u further locations each (had %llu)",
                                                static_cast<unsigned long long>(ng * attempts), ws.extra_cap, static_cast<unsigned long long>(before));
                    continue;
                }
                if (loc->lut_ext_depth >= depth_cap) fail(LCTY_ERR_RUNTIME, "window depth beyond 2 * reads + 2");
                ensure_depth_table(loc, std::min<uint64_t>(4ull * loc->lut_ext_depth, depth_cap));
            }
            after_batch(g0, ng, liks.data());
        }
    }
};

}  // namespace

namespace {

// one stage on the context's stream (lane 0) or on its side stream (lane 1)
void solve_stage_on(uint32_t lane, lcty_reads* reads, const uint16_t* genotypes, uint64_t n_gt, uint32_t ploidy, const double* priors,
                    const lcty_solver* solver, uint32_t attempts, const uint64_t* chain_seeds, double* lik_mean, double* lik_var,
                    double* liks_out) {
    if (!lik_mean || !lik_var) fail(LCTY_ERR_INVALID_INPUT, "null argument");
    StageRunner R(reads, genotypes, n_gt, ploidy, solver, attempts, chain_seeds, lane);
    R.run(genotypes, priors, chain_seeds, [&](uint64_t g0, uint64_t ng, const double* liks) {
        for (uint64_t g = 0; g < ng; g++) {
            const double* l = liks + g * attempts;
            math::mean_variance_or_nan(l, attempts, &lik_mean[g0 + g], &lik_var[g0 + g]);
            if (liks_out) memcpy(liks_out + (g0 + g) * attempts, l, sizeof(double) * attempts);
        }
    });
}

}  // namespace

// ---- RowGatherer (lcty_objects.hpp)
RowGatherer::RowGatherer(lcty_reads* owner_, const uint16_t* genotypes, uint64_t n_gt, uint32_t ploidy) : owner(owner_) {
    if (!owner || !genotypes) fail(LCTY_ERR_INVALID_INPUT, "null argument");
    ctx = owner->ctx; ctx->activate(); stream = ctx->stream;
    const uint32_t A = owner->locus->n_alleles;
    row_of.assign(A, 0xFFFF);
    for (uint64_t i = 0; i < n_gt * ploidy; i++) {
        if (genotypes[i] >= A) fail(LCTY_ERR_INVALID_INPUT, "genotype refers to allele %u >= %u", genotypes[i], A);
        row_of[genotypes[i]] = 0;
    }
    for (uint32_t a = 0; a < A; a++)
        if (row_of[a] == 0) { row_of[a] = static_cast<uint16_t>(alleles.size()); alleles.push_back(static_cast<uint16_t>(a)); }
    n_rows = static_cast<uint32_t>(alleles.size());
    if (n_rows == 0) fail(LCTY_ERR_INVALID_INPUT, "a stage without genotypes");
    auto& B = owner->gather;
    B.alleles.ensure(n_rows); B.alleles.upload(alleles.data(), n_rows, stream);
    B.row_of.ensure(A); B.row_of.upload(row_of.data(), A, stream);
}

void RowGatherer::count(lcty_reads* shard, uint32_t slot, uint64_t* good_out, uint64_t* extras_out) {
    if (!shard || shard->ctx != ctx || shard->locus != owner->locus) fail(LCTY_ERR_INVALID_INPUT, "the shards of a locus belong to one context and one locus");
    if (!shard->scored) fail(LCTY_ERR_INVALID_INPUT, "lcty_score_reads has not been called on this batch");
    shard->check_device_error();
    ensure_solver_tables(shard);
    auto& B = owner->gather;
    if (B.counters.n < 2ull * (slot + 1)) {
        // grow keeping what the earlier shards counted
        std::vector<unsigned long long> old(B.counters.n, 0ull);
        if (B.counters.n) { B.counters.download(old.data(), old.size(), stream); LCTY_HIP(hipStreamSynchronize(stream)); }
        old.resize(2ull * std::max<uint32_t>(slot + 1, 8), 0ull);
        B.counters.alloc(old.size()); B.counters.upload(old.data(), old.size(), stream);
    }
    unsigned long long zero[2] = {0, 0};
    LCTY_HIP(hipMemcpyAsync(B.counters.p + 2 * slot, zero, sizeof(zero), hipMemcpyHostToDevice, stream));
    const uint64_t n = static_cast<uint64_t>(n_rows) * shard->n_good_cached;
    if (n)
        hipLaunchKernelGGL(pack_rows_count_kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, stream,
                           reinterpret_cast<const LocCell*>(shard->d_loc_table.p), shard->ngp, static_cast<uint32_t>(shard->n_good_cached),
                           B.alleles.p, n_rows, B.counters.p + 2 * slot);
    LCTY_HIP(hipGetLastError());
    unsigned long long v = 0;
    LCTY_HIP(hipMemcpyAsync(&v, B.counters.p + 2 * slot, sizeof(v), hipMemcpyDeviceToHost, stream));
    LCTY_HIP(hipStreamSynchronize(stream));
    *good_out = shard->n_good_cached; *extras_out = v;
}

void RowGatherer::plan(const uint64_t* goods_, const uint64_t* extras, uint32_t n_shards_) {
    n_shards = n_shards_;
    goods.assign(goods_, goods_ + n_shards); first.assign(n_shards, 0);
    n_good = 0; stride = 0; ext_stride = 0;
    for (uint32_t r = 0; r < n_shards; r++) {
        first[r] = n_good; n_good += goods[r];
        stride = std::max(stride, goods[r]); ext_stride = std::max(ext_stride, extras[r]);
    }
    stride = std::max<uint64_t>(stride, 1); ext_stride = std::max<uint64_t>(ext_stride, 1);
    if (n_good >= (1ull << 24)) fail(LCTY_ERR_UNSUPPORTED, "the device solver handles up to 2^24 good read pairs per locus");
    if (ext_stride * n_shards > 0xFFFFFFFFull) fail(LCTY_ERR_UNSUPPORTED, "2^32 or more further pair-alignments over the shards of a locus");
    ngp = std::max<uint64_t>(64, (n_good + 63) / 64 * 64);
    // rows travel in chunks so that the staging buffers stay small next to the table (lcty_ctx_set_knob "gather_chunk_mb")
    const uint64_t chunk_bytes = static_cast<uint64_t>(ctx->knob("gather_chunk_mb", 256)) << 20;
    rows_per_chunk = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(n_rows, chunk_bytes / (stride * n_shards * sizeof(LocEntry)))));
    auto& B = owner->gather;
    size_t free_b = 0, total_b = 0;
    LCTY_HIP(hipMemGetInfo(&free_b, &total_b));
    const uint64_t need = static_cast<uint64_t>(n_rows) * ngp * sizeof(LocCell);
    if (B.table.n < need && need + need / 4 > free_b + B.table.n)
        fail(LCTY_ERR_RUNTIME, "device memory: the rows of %u alleles over %llu good read pairs (%.1f GB) do not fit", n_rows,
             static_cast<unsigned long long>(n_good), static_cast<double>(need) * 1e-9);
    B.table.ensure(need); B.ext.ensure(static_cast<uint64_t>(n_rows) * ngp); B.unm.ensure(ngp);
    B.send.ensure(chunk_cells() * sizeof(LocEntry)); B.recv.ensure(chunk_cells() * n_shards * sizeof(LocEntry));
    B.pa.ensure(ext_stride * n_shards); B.send_pa.ensure(ext_stride);
}

void RowGatherer::pack_chunk(lcty_reads* shard, uint32_t slot, uint32_t row0, uint8_t* cells, PairAlnDev* run) {
    const uint32_t nr = std::min(rows_per_chunk, n_rows - row0);
    const uint64_t n = static_cast<uint64_t>(nr) * stride;
    auto& B = owner->gather;
    hipLaunchKernelGGL(pack_rows_kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const LocCell*>(shard->d_loc_table.p), shard->d_loc_ext.p, shard->d_loc_unm.p, shard->ngp,
                       static_cast<uint32_t>(shard->n_good_cached), B.alleles.p + row0, nr, shard->d_pa.p, reinterpret_cast<LocEntry*>(cells), stride, run, B.counters.p + 2 * slot + 1);
    LCTY_HIP(hipGetLastError());
}

void RowGatherer::place_chunk(const uint8_t* cells, uint32_t shard, uint32_t row0) {
    const uint32_t nr = std::min(rows_per_chunk, n_rows - row0);
    const uint64_t n = static_cast<uint64_t>(nr) * goods[shard];
    if (!n) return;
    hipLaunchKernelGGL(place_rows_kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const LocEntry*>(cells), stride, static_cast<uint32_t>(goods[shard]), nr,
                       static_cast<uint32_t>(shard * ext_stride), reinterpret_cast<LocCell*>(owner->gather.table.p) + static_cast<size_t>(row0) * ngp,
                       owner->gather.ext.p + static_cast<size_t>(row0) * ngp, owner->gather.unm.p, row0 == 0, ngp, first[shard]);
    LCTY_HIP(hipGetLastError());
}

void RowGatherer::finish() {
    const uint64_t n = static_cast<uint64_t>(n_rows) * (ngp - n_good);
    if (n)
        hipLaunchKernelGGL(pad_rows_kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, stream,
                           reinterpret_cast<LocCell*>(owner->gather.table.p), owner->gather.ext.p, owner->gather.unm.p, ngp, n_good, n_rows);
    LCTY_HIP(hipGetLastError());
}

void lcty::solve_stage_gathered(lcty_reads* owner, const RowGatherer& G, const uint16_t* genotypes, uint64_t n_gt, uint32_t ploidy,
                          const double* priors, const lcty_solver* solver, uint32_t attempts, const uint64_t* chain_seeds,
                          double* lik_mean, double* lik_var, double* liks_out) {
    if (!lik_mean || !lik_var) fail(LCTY_ERR_INVALID_INPUT, "null argument");
    StageRunner R(owner, genotypes, n_gt, ploidy, solver, attempts, chain_seeds, 0, &G);
    R.run(genotypes, priors, chain_seeds, [&](uint64_t g0, uint64_t ng, const double* liks) {
        for (uint64_t g = 0; g < ng; g++) {
            const double* l = liks + g * attempts;
            math::mean_variance_or_nan(l, attempts, &lik_mean[g0 + g], &lik_var[g0 + g]);
            if (liks_out) memcpy(liks_out + (g0 + g) * attempts, l, sizeof(double) * attempts);
        }
    });
}

namespace {

uint32_t count_unexplained_on(hipStream_t s, lcty_reads* reads, const uint16_t* genotype, uint32_t ploidy) {
    lcty_ctx* ctx = reads->ctx;
    ctx->activate();
    reads->check_device_error();
    const uint32_t A = reads->locus->n_alleles;
    for (uint32_t i = 0; i < ploidy; i++)
        if (genotype[i] >= A) fail(LCTY_ERR_INVALID_INPUT, "genotype refers to allele %u >= %u", genotype[i], A);
    // buffers of the batch, grow-only: allocating or freeing here would wait for every stream of the device (the next locus' kernels)
    DevBuf<uint16_t>& d_ids = reads->d_unexpl_ids; DevBuf<unsigned long long>& d_out = reads->d_unexpl_out;
    d_ids.ensure(ploidy); d_ids.upload(genotype, ploidy, s);
    d_out.ensure(1); d_out.zero(s);
    const uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>((reads->n_pairs + 255) / 256, 4096));
    if (reads->n_pairs)
        hipLaunchKernelGGL(count_unexplained_kernel, dim3(blocks), dim3(256), 0, s, reads->d_status.p, reads->d_unmapped.p,
                           reads->d_matrix.p, reads->n_pairs, A, d_ids.p, ploidy, d_out.p);
    LCTY_HIP(hipGetLastError());
    unsigned long long v = 0;
    d_out.download(&v, 1, s);
    LCTY_HIP(hipStreamSynchronize(s));
    return static_cast<uint32_t>(v);
}

}  // namespace

extern "C" {


// read-back of the extended depth table of the solver stages (DistrCache past the LinearCache, distr_cache.rs:61-92 with
// bayes.rs:27-35 on the device): out[gc * width + depth], width = *width_io rounded up to a power of two >= 256
int32_t lcty_locus_depth_table(lcty_locus* locus, uint32_t* width_io, double* out) {
    return guarded([&] {
        if (!locus || !width_io) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        uint32_t w = LCTY_DEPTH_CACHE;
        while (w < *width_io) w *= 2;
        if (w > (1u << 22)) fail(LCTY_ERR_INVALID_INPUT, "depth table width %u", *width_io);
        *width_io = w;
        if (!out) return;
        lcty_ctx* ctx = locus->ctx;
        ctx->activate();
        ensure_depth_table(locus, w);
        // the locus may hold a wider table already: rows are lut_ext_depth apart
        std::vector<double> full(static_cast<size_t>(LCTY_GC_BINS) * locus->lut_ext_depth);
        locus->d_lut_ext.download(full.data(), full.size(), ctx->stream);
        LCTY_HIP(hipStreamSynchronize(ctx->stream));
        for (uint32_t g = 0; g < LCTY_GC_BINS; g++)
            memcpy(out + static_cast<size_t>(g) * w, full.data() + static_cast<size_t>(g) * locus->lut_ext_depth, sizeof(double) * w);
    });
}

// Greedy::default / SimAnneal::default (src/solvers/stoch.rs:45-52, 161-168)
int32_t lcty_solver_default(lcty_solver* s, int32_t kind) {
    return guarded([&] {
        if (!s || (kind != LCTY_SOLVER_GREEDY && kind != LCTY_SOLVER_ANNEAL && kind != LCTY_SOLVER_EXACT)) fail(LCTY_ERR_INVALID_INPUT, "unknown solver kind");
        memset(s, 0, sizeof(*s));
        s->kind = kind; s->best_start = 1; s->sample_size = 10;
        s->plato_size = kind == LCTY_SOLVER_GREEDY ? 100 : 10000;
        s->node_limit = kind == LCTY_SOLVER_EXACT ? 20u * 1000u * 1000u : 0u;
        s->anneal_steps = 20000; s->init_prob = kind == LCTY_SOLVER_EXACT ? 1e-4 : 0.5;      // exact: the relative gap at which the search stops = HiGHS' default mip_rel_gap, which the reference leaves alone (highs.rs:103-110); 0 = a proof of optimality
    });
}

// one chain seed per (genotype, attempt): consecutive next_u64() of Xoshiro256PlusPlus::seed_from_u64(master_seed)
int32_t lcty_chain_seeds(uint64_t master_seed, uint64_t n, uint64_t* out) {
    return guarded([&] {
        if (!out) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        uint64_t x = master_seed, s[4];
        for (int i = 0; i < 4; i++) {
            uint64_t z = (x += 0x9e3779b97f4a7c15ull);
            z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
            z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
            s[i] = z ^ (z >> 31);
        }
        auto rotl = [](uint64_t v, int k) { return (v << k) | (v >> (64 - k)); };
        for (uint64_t i = 0; i < n; i++) {
            out[i] = rotl(s[0] + s[3], 23) + s[0];
            const uint64_t t = s[1] << 17;
            s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        }
    });
}



int32_t lcty_solve_stage(lcty_reads* reads, const uint16_t* genotypes, uint64_t n_gt, uint32_t ploidy, const double* priors,
                         const lcty_solver* solver, uint32_t attempts, const uint64_t* chain_seeds,
                         double* lik_mean, double* lik_var, double* liks_out) {
    return guarded([&] { solve_stage_on(0, reads, genotypes, n_gt, ploidy, priors, solver, attempts, chain_seeds, lik_mean, lik_var, liks_out); });
}

// One stage over the reads of several batches of one locus held by ONE device (shards in read order): the rows of the stage's
// alleles are packed per shard and laid side by side exactly as lcty_solve_stage_read_sharded does between devices — the same
// code with the exchange left out; equals lcty_solve_stage on the unsharded batch bit for bit.
int32_t lcty_solve_stage_from_shards(lcty_reads* const* shards, uint32_t n_shards, const uint16_t* genotypes, uint64_t n_gt, uint32_t ploidy,
                                     const double* priors, const lcty_solver* solver, uint32_t attempts, const uint64_t* chain_seeds,
                                     double* lik_mean, double* lik_var, double* liks_out) {
    return guarded([&] {
        if (!shards || n_shards == 0 || !shards[0]) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        RowGatherer G(shards[0], genotypes, n_gt, ploidy);
        std::vector<uint64_t> goods(n_shards), extras(n_shards);
        for (uint32_t r = 0; r < n_shards; r++) G.count(shards[r], r, &goods[r], &extras[r]);
        G.plan(goods.data(), extras.data(), n_shards);
        for (uint32_t row0 = 0; row0 < G.n_rows; row0 += G.rows_per_chunk)
            for (uint32_t r = 0; r < n_shards; r++) {
                G.pack_chunk(shards[r], r, row0, G.recv_cells(r), G.run_of(r));
                G.place_chunk(G.recv_cells(r), r, row0);
            }
        G.finish();
        solve_stage_gathered(shards[0], G, genotypes, n_gt, ploidy, priors, solver, attempts, chain_seeds, lik_mean, lik_var, liks_out);
    });
}

int32_t lcty_assignment_counts(lcty_reads* reads, const uint16_t* genotype, uint32_t ploidy, const lcty_solver* solver,
                               uint32_t attempts, const uint64_t* chain_seeds, uint64_t* read_off, uint16_t* counts, uint64_t cap,
                               uint64_t* n_counts) {
    return guarded([&] {
        if (!read_off || !n_counts) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (attempts > 65535) fail(LCTY_ERR_UNSUPPORTED, "assignment counts are 16-bit (as in the reference, assgn.rs:94-96)");
        StageRunner R(reads, genotype, 1, ploidy, solver, attempts, chain_seeds);
        lcty_ctx* ctx = reads->ctx;
        hipStream_t s = ctx->stream;
        const uint64_t n_good = reads->n_good_cached;
        // GenotypeAlignments::read_ixs (assgn.rs:52-60): prefix sums of the number of locations of every read pair
        DevBuf<uint32_t> d_nw; d_nw.alloc(std::max<uint64_t>(n_good, 1));
        R.upload_genotypes(genotype, 1);
        if (n_good) {
            const uint32_t blocks = static_cast<uint32_t>((n_good + 255) / 256);
            switch (ploidy) {
                case 1: hipLaunchKernelGGL(read_nw_kernel<1>, dim3(blocks), dim3(256), 0, s, R.V, d_nw.p); break;
                case 2: hipLaunchKernelGGL(read_nw_kernel<2>, dim3(blocks), dim3(256), 0, s, R.V, d_nw.p); break;
                case 3: hipLaunchKernelGGL(read_nw_kernel<3>, dim3(blocks), dim3(256), 0, s, R.V, d_nw.p); break;
                default: hipLaunchKernelGGL(read_nw_kernel<4>, dim3(blocks), dim3(256), 0, s, R.V, d_nw.p); break;
            }
            LCTY_HIP(hipGetLastError());
        }
        std::vector<uint32_t> nw(n_good);
        d_nw.download(nw.data(), n_good, s);
        LCTY_HIP(hipStreamSynchronize(s));
        read_off[0] = 0;
        for (uint64_t g = 0; g < n_good; g++) read_off[g + 1] = read_off[g] + nw[g];
        *n_counts = read_off[n_good];
        if (!counts) return;
        if (cap < *n_counts) fail(LCTY_ERR_INVALID_INPUT, "counts buffer too small (%llu < %llu)", static_cast<unsigned long long>(cap),
                                  static_cast<unsigned long long>(*n_counts));
        DevBuf<uint64_t> d_off; d_off.alloc(n_good + 1); d_off.upload(read_off, n_good + 1, s);
        DevBuf<uint16_t> d_counts; d_counts.alloc(std::max<uint64_t>(*n_counts, 1)); d_counts.zero(s);
        R.run(genotype, nullptr, chain_seeds, [&](uint64_t, uint64_t, const double*) {
            // ReadAssignment::update_counts (assgn.rs:374-378) of every attempt, while the chains' lists are still on the device
            if (!n_good) return;
            const uint32_t blocks = static_cast<uint32_t>((n_good + 255) / 256);
            hipLaunchKernelGGL(assignment_counts_kernel, dim3(blocks), dim3(256), 0, s, R.V, d_nw.p, d_off.p, d_counts.p);
            LCTY_HIP(hipGetLastError());
            LCTY_HIP(hipStreamSynchronize(s));
        });
        d_counts.download(counts, *n_counts, s);
        LCTY_HIP(hipStreamSynchronize(s));
    });
}

int32_t lcty_count_unexplained(lcty_reads* reads, const uint16_t* genotype, uint32_t ploidy, uint32_t* out) {
    return guarded([&] {
        if (!reads || !genotype || !out || ploidy == 0) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (!reads->scored) fail(LCTY_ERR_INVALID_INPUT, "lcty_score_reads has not been called on this batch");
        *out = count_unexplained_on(reads->ctx->stream, reads, genotype, ploidy);
    });
}

// Genotyping::{find_weighted_dist, check_first_prob, check_num_of_reads} (solve.rs:621-675) with genotype_distance
// (339-357) over gen_permutations (ext/vec.rs:342-372: for three or more elements Heap's algorithm as written there never
// hands the unpermuted order to the callback, so it is not among the candidates — kept as is)
int32_t lcty_call_checks(const uint16_t* genotypes, uint64_t n, uint32_t ploidy, const double* ln_probs, uint32_t n_reads,
                         const uint32_t* dist, uint32_t n_alleles, uint32_t* distances_out, double* weighted_dist, uint32_t* warnings) {
    return guarded([&] {
        if (!genotypes || !ln_probs || n == 0 || ploidy == 0) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (ploidy > 8) fail(LCTY_ERR_UNSUPPORTED, "ploidy above 8");
        uint32_t w = 0;
        const double lp0 = ln_probs[0];
        if (std::isnan(lp0) || lp0 < -2.0 * 2.302585092994045684) w |= LCTY_WARN_NO_PROBABLE_GENOTYPE;     // < 0.01
        if (n_reads < ploidy) w |= LCTY_WARN_FEW_READS;
        else if (ploidy > 1 && n_reads < ploidy * 10) {
            const double k = ploidy, nr = n_reads;
            if (std::exp(std::log(k - 1.0) * nr - std::log(k) * (nr - 1.0)) > 0.1) w |= LCTY_WARN_FEW_READS;
        }
        if (warnings) *warnings = w;
        if (!dist) { if (weighted_dist) *weighted_dist = std::numeric_limits<double>::quiet_NaN(); return; }
        for (uint64_t i = 0; i < n * ploidy; i++)
            if (genotypes[i] >= n_alleles) fail(LCTY_ERR_INVALID_INPUT, "genotype refers to allele %u >= %u", genotypes[i], n_alleles);
        auto pair_dist = [&](const uint16_t* a, const uint16_t* b) -> uint32_t {        // one permutation of gt1 against gt2
            uint32_t d = 0;
            for (uint32_t t = 0; t < ploidy; t++) {
                if (a[t] == b[t]) continue;
                const uint32_t v = dist[static_cast<size_t>(a[t]) * n_alleles + b[t]];
                if (v == LCTY_NONE_U32) return LCTY_NONE_U32;
                d += v;
            }
            return d;
        };
        const uint16_t* g0 = genotypes;
        double sum_prob = 0.0, sum_dist = 0.0;
        bool all_known = true;
        for (uint64_t i = 0; i < n; i++) {
            const double prob = std::exp(ln_probs[i]);
            sum_prob += prob;
            uint32_t best = 0;
            if (i > 0) {
                const uint16_t* g = genotypes + i * ploidy;
                best = LCTY_NONE_U32;
                uint16_t buf[8];
                for (uint32_t t = 0; t < ploidy; t++) buf[t] = g0[t];
                if (ploidy == 1) best = pair_dist(buf, g);
                else if (ploidy == 2) {
                    best = pair_dist(buf, g);
                    std::swap(buf[0], buf[1]);
                    best = std::min(best, pair_dist(buf, g));
                } else {
                    uint32_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    for (uint32_t k = 1; k < ploidy;) {
                        if (c[k] < k) {
                            std::swap(buf[k], buf[(k & 1u) ? c[k] : 0u]);
                            best = std::min(best, pair_dist(buf, g));
                            c[k]++; k = 1;
                        } else { c[k] = 0; k++; }
                    }
                }
            }
            if (distances_out) distances_out[i] = best;
            if (best == LCTY_NONE_U32) all_known = false;
            else sum_dist += prob * static_cast<double>(best);
        }
        if (weighted_dist) *weighted_dist = all_known ? sum_dist / sum_prob : std::numeric_limits<double>::quiet_NaN();
    });
}

// Scheme::default (solve.rs:211-230)
int32_t lcty_stages_default(lcty_stage* stages, uint32_t* n_stages) {
    return guarded([&] {
        if (!stages || !n_stages) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        memset(stages, 0, 2 * sizeof(lcty_stage));
        lcty_solver_default(&stages[0].solver, LCTY_SOLVER_GREEDY); stages[0].in_size = 5000; stages[0].attempts = 1;
        lcty_solver_default(&stages[1].solver, LCTY_SOLVER_ANNEAL); stages[1].in_size = 20; stages[1].attempts = 20;
        *n_stages = 2;
    });
}

}  // extern "C"

namespace {

// solve::solve (solve.rs:926-981) with solve_single_thread (789-857) as the stage loop, in two halves: `head` = run_filter and every
// stage but the last, `tail` = the last stage (few genotypes, many attempts: long serial chains that leave most of the GPU idle),
// the final comparison and the checks. lcty_solve runs them back to back; lcty_solve_queue runs the tail of a locus on the context's
// side stream while the head of the next locus has the main one.
struct LocusRun {
    lcty_reads* reads = nullptr; uint32_t ploidy = 2; const lcty_stage* stages = nullptr; uint32_t n_stages = 0;
    uint64_t master_seed = 0; const double* priors = nullptr; lcty_call* out = nullptr;
    uint64_t G = 0, n = 0, threads = 1;
    std::vector<uint16_t> gts; std::vector<uint64_t> ixs; std::vector<double> mean, var; std::vector<uint32_t> att;

    static void ok(int32_t rc) { if (rc != LCTY_OK) throw Error(rc, std::string(lcty_last_error())); }
    // lcty_ctx_set_knob "queue_trace" 1: wall-clock marks of the phases of a locus on stderr (where does a step of the queue go?)
    void mark(const char* what) const {
        if (!reads || reads->ctx->knob("queue_trace", 0) == 0) return;
        const double t = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
        fprintf(stderr, "[lcty queue] %.3f ms batch %p %s\n", t, static_cast<const void*>(reads), what);
    }

    void stage(uint32_t si, uint32_t lane) {
        const bool last = si + 1 == n_stages;
        const lcty_params& prm = reads->locus->prm;
        const uint64_t out_size = last ? 0 : stages[si + 1].in_size;
        if (!(prm.dont_skip || last || out_size < n)) return;                    // "Skipping stage, not enough genotypes"
        const uint32_t attempts = stages[si].attempts;
        std::vector<uint16_t> sub(n * ploidy); std::vector<double> pri(n), m(n), v(n); std::vector<uint64_t> seeds(n * attempts);
        for (uint64_t t = 0; t < n; t++) {
            memcpy(sub.data() + t * ploidy, gts.data() + ixs[t] * ploidy, ploidy * sizeof(uint16_t));
            pri[t] = priors ? priors[ixs[t]] : 0.0;
        }
        ok(lcty_chain_seeds(master_seed + static_cast<uint64_t>(si + 1) * 0x9e3779b97f4a7c15ull, n * attempts, seeds.data()));
        mark(lane ? "tail stage: inputs ready" : "head stage: inputs ready");
        solve_stage_on(lane, reads, sub.data(), n, ploidy, pri.data(), &stages[si].solver, attempts, seeds.data(), m.data(), v.data(), nullptr);
        mark(lane ? "tail stage: chains done" : "head stage: chains done");
        for (uint64_t t = 0; t < n; t++) { mean[ixs[t]] = m[t]; var[ixs[t]] = v[t]; att[ixs[t]] = attempts; }
        if (!last) ok(lcty_discard_improbable(mean.data(), var.data(), att.data(), ixs.data(), n, prm.prob_thresh, out_size, threads, &n));
        mark(lane ? "tail stage: discarded" : "head stage: discarded");
    }

    void head(bool score) {
        if (!reads || !stages || !out || n_stages == 0) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        mark("head: start");
        if (score) ok(lcty_score_reads(reads));
        mark("head: scoring launched");
        if (!reads->scored) fail(LCTY_ERR_INVALID_INPUT, "lcty_score_reads has not been called on this batch");
        for (uint32_t s = 0; s < n_stages; s++)
            if (stages[s].attempts == 0 || stages[s].in_size == 0) fail(LCTY_ERR_INVALID_INPUT, "stage %u: attempts and in_size must be positive", s);
        const lcty_locus* loc = reads->locus;
        const lcty_params& prm = loc->prm;
        const uint32_t A = loc->n_alleles;
        G = count_genotypes(A, ploidy);
        gts.resize(G * ploidy);
        ok(lcty_generate_genotypes(A, ploidy, gts.data(), G));
        ixs.resize(G);
        std::iota(ixs.begin(), ixs.end(), 0ull);
        n = G;
        memset(out, 0, sizeof(*out));
        // filter (solve.rs:940-945): run_filter gets data.threads as the floor of kept genotypes; the stage loop passes ONE_THREAD
        // to discard_improbable_genotypes when threads == 1 (solve.rs:797, 853) and data.threads otherwise (1087-1089)
        threads = std::max<uint64_t>(1, prm.threads);
        if (prm.dont_skip || stages[0].in_size < G) {
            if (ploidy == 2) {
                // the scores stay on the device: sorted and cut there, only the kept indices come back (lcty_select.hip)
                ok(lcty_prefilter_async(reads, 2));
                if (priors) ok(lcty_prefilter_add_priors(reads, priors, G));
                ok(lcty_prefilter_truncate(reads, prm.filt_diff, stages[0].in_size, threads, ixs.data(), G, &n));
            } else {
                std::vector<double> scores(G);
                ok(lcty_prefilter(reads, nullptr, G, ploidy, priors, scores.data()));
                ok(lcty_truncate(scores.data(), ixs.data(), G, prm.filt_diff, stages[0].in_size, threads, &n));
            }
        }
        out->kept_after_filter = n;
        mark("head: prefiltered and truncated");
        mean.assign(G, std::numeric_limits<double>::quiet_NaN()); var.assign(G, std::numeric_limits<double>::quiet_NaN());
        att.assign(G, 0);
        for (uint32_t si = 0; si + 1 < n_stages; si++) stage(si, 0);
    }

    void tail(uint32_t lane) {
        lcty_ctx* ctx = reads->ctx;
        ctx->activate();
        const lcty_params& prm = reads->locus->prm;
        mark("tail: start");
        stage(n_stages - 1, lane);
        ok(lcty_produce_result(mean.data(), var.data(), att.data(), ixs.data(), n, prm.prob_thresh, 0, out->ixs, out->ln_probs, &out->n_out,
                               &out->quality));
        out->unexpl_reads = count_unexplained_on(lane ? ctx->side_stream() : ctx->stream, reads, gts.data() + out->ixs[0] * ploidy, ploidy);
        out->n_good = reads->n_good_cached;
        std::vector<uint16_t> res(out->n_out * ploidy);
        for (uint64_t t = 0; t < out->n_out; t++) memcpy(res.data() + t * ploidy, gts.data() + out->ixs[t] * ploidy, ploidy * sizeof(uint16_t));
        ok(lcty_call_checks(res.data(), out->n_out, ploidy, out->ln_probs, static_cast<uint32_t>(std::min<uint64_t>(out->n_good, 0xFFFFFFFFull)),
                            nullptr, reads->locus->n_alleles, nullptr, nullptr, &out->warnings));
    }
};

}  // namespace

extern "C" {

int32_t lcty_solve(lcty_reads* reads, uint32_t ploidy, const lcty_stage* stages, uint32_t n_stages, uint64_t master_seed,
                   const double* priors, lcty_call* out, double* lik_mean_out, double* lik_var_out, uint32_t* attempts_out) {
    return guarded([&] {
        LocusRun R;
        R.reads = reads; R.ploidy = ploidy; R.stages = stages; R.n_stages = n_stages; R.master_seed = master_seed; R.priors = priors; R.out = out;
        R.head(false);
        R.tail(0);
        if (lik_mean_out) memcpy(lik_mean_out, R.mean.data(), R.G * sizeof(double));
        if (lik_var_out) memcpy(lik_var_out, R.var.data(), R.G * sizeof(double));
        if (attempts_out) memcpy(attempts_out, R.att.data(), R.G * sizeof(uint32_t));
    });
}

}  // extern "C" (reopened below)

namespace {
// the queue of lcty_solve_queue / lcty_solve_queue_fed: `batch_of(i)` right before position i is scored, `done_with(i)` once its
// last stage has been joined
template <typename GET, typename DONE>
void run_queue(uint32_t n, GET&& batch_of, DONE&& done_with, uint32_t ploidy, const lcty_stage* stages, uint32_t n_stages,
               const uint64_t* master_seeds, const double* const* priors, lcty_call* out) {
    std::unique_ptr<LocusRun> prev;
    std::thread tail_thread;
    int32_t tail_rc = LCTY_OK; std::string tail_msg;
    lcty_ctx* ctx = nullptr;
    uint32_t tail_of = 0;
    auto release_gate = [&] {
        if (!ctx) return;
        { std::lock_guard<std::mutex> lock(ctx->gate.m); if (ctx->gate.target > ctx->gate.epoch) ctx->gate.epoch = ctx->gate.target; }
        ctx->gate.cv.notify_all();
    };
    auto join_tail = [&] {
        const bool had = tail_thread.joinable();
        if (had) tail_thread.join();
        prev.reset();
        if (tail_rc != LCTY_OK) { const int32_t rc = tail_rc; tail_rc = LCTY_OK; fail(rc, "%s", tail_msg.c_str()); }
        if (had) done_with(tail_of);
    };
    try {
        for (uint32_t i = 0; i < n; i++) {
            auto R = std::make_unique<LocusRun>();
            R->reads = batch_of(i);
            ctx = R->reads->ctx;
            R->ploidy = ploidy; R->stages = stages; R->n_stages = n_stages; R->master_seed = master_seeds[i];
            R->priors = priors ? priors[i] : nullptr; R->out = &out[i];
            try { R->head(true); }
            catch (...) { release_gate(); throw; }
            release_gate();                                                  // a head that launched no greedy loop must not keep the tail waiting
            join_tail();
            prev = std::move(R);
            LocusRun* run = prev.get();
            {
                // the tail of this locus lets the greedy loop of the next locus go first (lcty_ctx::LaunchGate)
                std::lock_guard<std::mutex> lock(ctx->gate.m);
                ctx->gate.target = i + 1 < n && n_stages > 1 && stages[0].solver.kind == LCTY_SOLVER_GREEDY ? ctx->gate.epoch + 1 : 0;
            }
            tail_of = i;
            tail_thread = std::thread([run, &tail_rc, &tail_msg] {
                try { run->tail(1); }
                catch (const Error& e) { tail_rc = e.code; tail_msg = e.what(); }
                catch (const std::exception& e) { tail_rc = LCTY_ERR_RUNTIME; tail_msg = e.what(); }
            });
        }
        join_tail();
    } catch (...) {
        release_gate();
        if (tail_thread.joinable()) tail_thread.join();
        throw;
    }
}
}  // namespace

extern "C" {

// The genotyping loop of `locityper genotype` over its loci (genotype.rs:1331-1351: analyze_locus one after the other) as a queue
// on one GPU. Each entry is a batch of read pairs of its own locus, appended but not necessarily scored: for every entry
// lcty_score_reads + lcty_solve. The loci are independent, so the last stage of locus i (the annealing attempts: a few hundred
// serial chains that occupy a few per cent of the device) runs on the context's side stream from a second host thread while
// locus i + 1 is scored, prefiltered and greedily solved on the main stream. Results are those of lcty_solve entry by entry
// (a chain's random stream is its seed). An entry may appear again later in the queue, not next to itself; neighbours must
// belong to different loci (lcty_locus objects): a stage may rebuild its locus' depth table.
int32_t lcty_solve_queue(lcty_reads* const* batches, uint32_t n_batches, uint32_t ploidy, const lcty_stage* stages, uint32_t n_stages,
                         const uint64_t* master_seeds, const double* const* priors, lcty_call* out) {
    return guarded([&] {
        if (!batches || !stages || !master_seeds || !out || n_stages == 0) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        for (uint32_t i = 0; i < n_batches; i++) {
            if (!batches[i]) fail(LCTY_ERR_INVALID_INPUT, "null batch");
            if (batches[i]->ctx != batches[0]->ctx) fail(LCTY_ERR_INVALID_INPUT, "the batches of a queue share one context");
            if (i && (batches[i] == batches[i - 1] || batches[i]->locus == batches[i - 1]->locus))
                fail(LCTY_ERR_INVALID_INPUT, "neighbours in the queue must be different batches of different loci");
        }
        run_queue(n_batches, [&](uint32_t i) { return batches[i]; }, [](uint32_t) {}, ploidy, stages, n_stages, master_seeds, priors, out);
    });
}

// The same queue with the batches handed over one at a time: `acquire(user, i)` is called right before position i is scored and
// returns its batch — filled by then, e.g. by a host thread that appends the chunks of locus i while locus i - 1 is being solved
// (the appends have a stream of their own) —, `release(user, i)` when the last stage of position i has finished and nothing of
// the batch is in use any more (lcty_reads_reset may then bind it to another locus). Position i is released before position
// i + 2 is acquired: three batch objects carry a queue of any length. The loop of `locityper genotype` over its loci
// (genotype.rs:1331-1351) with the loading of locus i + 1 next to the solving of locus i.
int32_t lcty_solve_queue_fed(uint32_t n_loci, lcty_queue_acquire_fn acquire, lcty_queue_release_fn release, void* user, uint32_t ploidy,
                             const lcty_stage* stages, uint32_t n_stages, const uint64_t* master_seeds, const double* const* priors, lcty_call* out) {
    return guarded([&] {
        if (!acquire || !stages || !master_seeds || !out || n_stages == 0) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        lcty_reads* before = nullptr;
        run_queue(n_loci, [&](uint32_t i) {
            lcty_reads* r = acquire(user, i);
            if (!r) fail(LCTY_ERR_INVALID_INPUT, "the queue's source has no batch for position %u", i);
            if (before && (r == before || r->locus == before->locus || r->ctx != before->ctx))
                fail(LCTY_ERR_INVALID_INPUT, "neighbours in the queue must be different batches of different loci in one context");
            before = r;
            return r;
        }, [&](uint32_t i) { if (release) release(user, i); }, ploidy, stages, n_stages, master_seeds, priors, out);
    });
}

int32_t lcty_solve_stats(const lcty_reads* reads, uint64_t* chains, uint64_t* iterations, uint64_t* accepted) {
    return guarded([&] {
        if (!reads) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (chains) *chains = reads->stat_chains;
        if (iterations) *iterations = reads->stat_iterations;
        if (accepted) *accepted = reads->stat_accepted;
    });
}

// discard_improbable_genotypes (src/solvers/solve.rs:425-480): ixs in/out, *n_keep = new count
int32_t lcty_discard_improbable(const double* lik_mean, const double* lik_var, const uint32_t* attempts, uint64_t* ixs, uint64_t n,
                                double prob_thresh, uint64_t out_size, uint64_t threads, uint64_t* n_keep) {
    return guarded([&] {
        if (!lik_mean || !lik_var || !attempts || !ixs || !n_keep) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        out_size = std::max(out_size, threads);
        if (prob_thresh == -std::numeric_limits<double>::infinity() || out_size >= n) { *n_keep = n; return; }
        sort_by_mean(lik_mean, ixs, n);
        const uint64_t best = ixs[0];
        uint64_t m = out_size;
        if (out_size <= 500) {                          // SOPHISTICATED_COUNT
            uint32_t dropped = 0;
            for (uint64_t t = out_size; t < n; t++) {
                const uint64_t ix = ixs[t];
                const double ln_pval = compare_two(lik_mean[ix], lik_var[ix], attempts[ix], lik_mean[best], lik_var[best], attempts[best]);
                if (ln_pval >= prob_thresh) ixs[m++] = ix;
                else if (++dropped >= 5) break;         // STOP_COUNT
            }
        }
        *n_keep = m;
    });
}

// produce_result (src/solvers/solve.rs:482-535): out arrays sized >= min(n, 50)
int32_t lcty_produce_result(const double* lik_mean, const double* lik_var, const uint32_t* attempts, const uint64_t* ixs_in, uint64_t n_in,
                            double prob_thresh, uint64_t out_bams, uint64_t* out_ixs, double* out_ln_probs, uint64_t* n_out,
                            double* quality) {
    return guarded([&] {
        if (!lik_mean || !lik_var || !attempts || !ixs_in || !out_ixs || !out_ln_probs || !n_out) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (n_in == 0) fail(LCTY_ERR_INVALID_INPUT, "no genotypes");
        const double THRESH = -11.512925464970229;
        const uint64_t min_output = std::max<uint64_t>(4, out_bams);
        const double thresh_prob = std::fmin(THRESH, prob_thresh);
        std::vector<uint64_t> ixs(ixs_in, ixs_in + n_in);
        sort_by_mean(lik_mean, ixs.data(), n_in);
        uint64_t n = std::min<uint64_t>(n_in, 50);      // MAX_GENOTYPES
        std::vector<double> ln_probs(n, 0.0);
        for (uint64_t i = 0; i < n; i++) {
            const uint64_t u = ixs[i];
            for (uint64_t j = i + 1; j < n; j++) {
                const uint64_t v = ixs[j];
                const double prob_j = compare_two(lik_mean[v], lik_var[v], attempts[v], lik_mean[u], lik_var[u], attempts[u]);
                if (i == 0 && j >= min_output && prob_j < thresh_prob) { n = j; break; }
                ln_probs[i] += std::log1p(-std::exp(prob_j));
                ln_probs[j] += prob_j;
            }
        }
        const double norm = ln_sum(ln_probs.data(), n);
        for (uint64_t t = 0; t < n; t++) { out_ixs[t] = ixs[t]; out_ln_probs[t] = ln_probs[t] - norm; }
        *n_out = n;
        if (quality) {
            std::vector<double> rest(out_ln_probs + (n ? 1 : 0), out_ln_probs + n);
            *quality = std::fmin(-10.0 * (ln_sum(rest.data(), rest.size()) * 0.4342944819032518277), 1e9);   // Phred::from_ln_prob
        }
    });
}

}  // extern "C"
