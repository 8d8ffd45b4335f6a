// lcty_solve_kernels.hip — the solver stages of `locityper genotype` on gfx950.
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
#include <cmath>
#include <type_traits>
#include <vector>

#include "lcty_solve_device.hpp"

namespace lcty {

// Forms of the greedy loop (template parameter of the kernel; lcty_ctx_set_knob "solve_greedy_form" picks among the compiled ones):
//   32  (diagnostic) the phases of an iteration timed with the shader clock, summed per wavefront into SolveView::dbg
// Tried in round 4 and dropped (profiles/r04_greedy_forms_and_phases.txt): duplicate check / row maximum through LDS, the depth table
// as pairs, 16-byte record loads, a lane per alternative (further alternatives of a read on spare lanes of its row) — none moved the
// loop. What an iteration costs is the cache lines it has to bring into the CU — one record line from HBM and four table lines from
// the L2 per candidate, ~250 per wavefront-iteration, ~165 in flight per CU — whatever the instructions around them look like.

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

// ---------------- K12 + K13 of a diploid stage: one 256-thread workgroup per GROUP of chains (InitGroup) ----------------
// solve_init_kernel streams two rows of the location table per chain: 40 B read + 32 B written per (chain, read), and with 5 000
// chains out of a few hundred alleles every row is read forty times over. Here a workgroup builds the records of up to INIT_TILE_T
// chains whose genotypes lie on at most INIT_TILE_R rows: wavefront k takes the k-th range of the locus' reads as before, loads the
// rows' cells of a block of 64 reads ONCE (next block's in flight), parks them in its own LDS strip (a lane reads back only what it
// wrote: the strip is a register file indexed by a uniform number) and walks the group's chains over them. Per (chain, read):
// 32 B written, 16 B x rows / chains + 8 B / chains read.
// The body per chain is straight-line for what nearly every read is — at most one pair-alignment within the threshold on either
// contig, so its locations are a subset of {contig 0, contig 1, both unmapped}: order by three comparisons, the tweak hash of a
// location from a per-read base shared by the group's chains, no loop. A read with several pair-alignments on a contig in reach
// (~1 %) gets its place in the record list like the others (whether it is non-trivial is looked up at once in the rare case it is
// not evident) and is put on the wavefront's list of deferred reads; the list is worked off 64 entries at a time by the general
// code (locs_from_cells + LocIter, one entry per lane, chain parameters per lane) between two blocks.
__device__ __forceinline__ uint64_t counter_fin(uint64_t z) {                 // the finaliser of counter_u64
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
constexpr uint64_t GOLDEN64 = 0x9e3779b97f4a7c15ull;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_16(uint4* p, uint4 v) {              // one 16-byte store that stays one
    u32x4_t x; x.x = v.x; x.y = v.y; x.z = v.z; x.w = v.w;
    __builtin_nontemporal_store(x, reinterpret_cast<u32x4_t*>(p));
}
__host__ __device__ inline size_t init_tile_lds(uint32_t T, uint32_t R, uint32_t wstride) {
    const size_t depth = (static_cast<size_t>(T) * wstride * 4 + 15) & ~static_cast<size_t>(15);
    return depth + static_cast<size_t>(T) * 256 * 8 + 4ull * R * 64 * 16 + static_cast<size_t>(T) * 4 * 8 + static_cast<size_t>(T) * 4 * 5 + 64;
}

// the general code for the deferred reads of a wavefront: lane i takes the i-th of the first n entries of its list (a ring in global
// memory, written and read by this wavefront only; read behind a workgroup-scope release fence). What the reads add to the chains' alignment likelihood is summed
// per chain in the order the reads were listed — increasing read number — whatever else shares the list: c_aln must not depend on
// how the stage was cut into groups.
__device__ __forceinline__ void init_tile_deferred(const SolveView& V, const InitChainP* __restrict__ cp, uint32_t T, uint32_t* depth, double* aln_def,
                                                   uint32_t* ex_cnt, unsigned long long* queue, uint32_t head, uint32_t n, uint32_t wave, uint32_t lane, bool random_start) {
    double add = 0.0;
    uint32_t my_c = 0xFFFFFFFFu;
    if (lane < n) {
        const unsigned long long packed = __hip_atomic_load(&queue[(head + lane) & (INIT_TILE_Q - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint2 item = make_uint2(static_cast<uint32_t>(packed), static_cast<uint32_t>(packed >> 32));
        const uint32_t c = item.x >> 24, rp = item.x & 0xFFFFFFu, slot = item.y;
        const InitChainP P = cp[c];
        Geno<2> G;
        G.id[0] = P.id0; G.id[1] = P.id1; G.row[0] = P.row0; G.row[1] = P.row1; G.shift[0] = P.shift0; G.shift[1] = P.shift1;
        G.reg_start[0] = P.rs0; G.reg_start[1] = P.rs1; G.reg_end[0] = P.re0; G.reg_end[1] = P.re1; G.total_w = P.total_w;
        Locs<2> L;
        locs_init<2>(L, V, rp, G);
        if (L.nw > 255) atomicMax(V.overflow, 2u);                               // a record keeps the location in 8 bits
        uint32_t a0 = 0;
        if (L.nw > 1 && random_start)
            a0 = static_cast<uint32_t>(__umul64hi(counter_u64(P.seed ^ INIT_KEY_XOR, rp), static_cast<uint64_t>(L.nw)));
        const uint32_t nloc = min(L.nw, 255u), n_extra = nloc > 2 ? nloc - 2u : 0u;
        uint32_t eix = 0;
        if (n_extra) eix = atomicAdd(&ex_cnt[c], n_extra);
        const bool room = static_cast<uint64_t>(eix) + n_extra <= V.extra_cap;
        ExtraLoc* extra = V.extra + static_cast<uint64_t>(P.chain) * V.extra_cap;
        ChainRec rec; rec.rp_cur = rp | (a0 << 24); rec.meta = nloc | (eix << 8);
        rec.lp0 = rec.lp1 = 0.0; rec.win0 = rec.win1 = 0;
        LocIter<2> it; it.start(L);
        LocOut o;
        for (uint32_t t = 0; t < nloc && it.next(L, V, o); t++) {
            uint32_t wa, wb;
            loc_windows(V, G, o, P.seed, rp, t, &wa, &wb);
            const uint32_t win = wa | (wb << 16);
            if (t == 0) { rec.lp0 = o.lp; rec.win0 = win; }
            else if (t == 1) { rec.lp1 = o.lp; rec.win1 = win; }
            else if (room) { ExtraLoc e; e.lp = o.lp; e.win = win; e._pad = 0; extra[eix + t - 2] = e; }
            if (t == a0) {
                atomicAdd(&depth[static_cast<size_t>(c) * V.wstride + wa], 1u);
                atomicAdd(&depth[static_cast<size_t>(c) * V.wstride + wb], 1u);
                add = o.lp;
                my_c = c;
            }
        }
        if (L.nw > 1) V.recs[static_cast<uint64_t>(P.chain) * V.rstride + static_cast<uint64_t>(wave) * V.seg_reads + slot] = rec;
    }
    for (uint32_t c = 0; c < T; c++) {
        unsigned long long m = __ballot(my_c == c);
        if (!m) continue;
        double acc = aln_def[c * 4 + wave];
        while (m) {
            const int l = __ffsll(static_cast<long long>(m)) - 1;
            acc += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(add), l), __builtin_amdgcn_readlane(__double2loint(add), l));
            m &= m - 1;
        }
        if (lane == 0) aln_def[c * 4 + wave] = acc;
    }
}

__global__ __launch_bounds__(256) void solve_init_tile_kernel(const SolveView V, const InitGroup* __restrict__ groups, const InitChainP* __restrict__ chains,
                                                             unsigned long long* lists, const uint32_t T_max, const uint32_t R_max) {
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const InitGroup& grp = groups[blockIdx.x];
    const uint32_t T = grp.n_chains, R = grp.n_rows;
    const InitChainP* __restrict__ cp = chains + grp.first;
    const uint32_t W = V.wstride;
    uint32_t* depth = reinterpret_cast<uint32_t*>(smem);                            // [T_max][W]
    size_t at = (static_cast<size_t>(T_max) * W * 4 + 15) & ~static_cast<size_t>(15);
    double* aln = reinterpret_cast<double*>(smem + at); at += static_cast<size_t>(T_max) * 256 * 8;      // [T_max][256] every thread's share of c_aln
    uint4* stage = reinterpret_cast<uint4*>(smem + at) + static_cast<size_t>(wave) * R_max * 64; at += 4ull * R_max * 64 * 16;   // [4][R_max][64]
    unsigned long long* queue = lists + (static_cast<size_t>(blockIdx.x) * 4 + wave) * INIT_TILE_Q;     // this wavefront's list of deferred reads
    double* aln_def = reinterpret_cast<double*>(smem + at); at += static_cast<size_t>(T_max) * 4 * 8;      // [T_max][4] the deferred reads' share of c_aln, per wavefront
    uint32_t* ex_cnt = reinterpret_cast<uint32_t*>(smem + at); at += static_cast<size_t>(T_max) * 4;    // further locations handed out per chain
    uint32_t* seg_cnt = reinterpret_cast<uint32_t*>(smem + at);                     // [T_max][4] records of every segment

    // K12: window distributions of every chain of the group (apply_tweak, assgn.rs:140-150)
    for (uint32_t c = 0; c < T; c++) {
        const InitChainP P = cp[c];
        double* ww = V.c_ww + static_cast<uint64_t>(P.chain) * W;
        uint8_t* wgc = V.c_gc + static_cast<uint64_t>(P.chain) * W;
        uint32_t* wuc = V.n_wk ? V.c_uc + static_cast<uint64_t>(P.chain) * W : nullptr;
        for (uint32_t w = tid; w < P.total_w; w += 256) {
            depth[c * W + w] = 0;
            double weight = 0.0; uint32_t g = 0, uc = V.n_wk - 1;                  // windows 0 and 1 (unmapped / out of region): trivial
            if (w >= 2) {
                const bool second = w >= P.shift1;
                const uint32_t allele = second ? P.id1 : P.id0, sh = second ? P.shift1 : P.shift0, rs = second ? P.rs1 : P.rs0;
                const uint32_t start = rs + (w - sh) * V.window, end = start + V.window;
                const uint32_t left = min(V.tweak, start), right = min(V.tweak, V.allele_len[allele] - end);
                const uint64_t r = counter_u64(P.seed ^ WINDOW_KEY_XOR, w);               // rng.random_range(-left..=right)
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
        aln[c * 256 + tid] = 0.0;
        if (tid < 4) aln_def[c * 4 + tid] = 0.0;
        if (tid == 0) ex_cnt[c] = 0;
    }
    __syncthreads();

    // K13: initial assignment, depth histograms, the records of the non-trivial reads. The wavefronts do not talk to each other.
    const bool random_start = V.solver.kind == LCTY_SOLVER_ANNEAL || (V.solver.kind == LCTY_SOLVER_GREEDY && !V.solver.best_start);
    const uint32_t seg_lo = min(wave * V.seg_reads, V.n_good), seg_hi = min(seg_lo + V.seg_reads, V.n_good);
    uint64_t row_at[INIT_TILE_R];
#pragma unroll
    for (uint32_t r = 0; r < INIT_TILE_R; r++) row_at[r] = static_cast<uint64_t>(grp.row[r < R ? r : 0]) * V.ngp;
    const uint4* __restrict__ table = reinterpret_cast<const uint4*>(V.table);      // LocCell: {lp lo, lp hi, m1n, m2}
    uint4 nxt[INIT_TILE_R]; double nxt_unm = 0.0;
#pragma unroll
    for (uint32_t r = 0; r < INIT_TILE_R; r++) nxt[r] = make_uint4(0, 0, 0, 0);
    if (seg_lo + lane < seg_hi) {
#pragma unroll
        for (uint32_t r = 0; r < INIT_TILE_R; r++) if (r < R) nxt[r] = table[row_at[r] + seg_lo + lane];
        nxt_unm = V.table_unm[seg_lo + lane];
    }
    uint32_t n_recs_all = 0;                   // lane c: the records of chain c so far (read and written with a uniform lane number)
    uint32_t qn = 0, qhead = 0;                // entries on the list of deferred reads (a ring), its first
    for (uint32_t base = seg_lo; base < seg_hi; base += 64) {
        const uint32_t rp = base + lane;
        const bool in = rp < seg_hi;
#pragma unroll
        for (uint32_t r = 0; r < INIT_TILE_R; r++) if (r < R) stage[r * 64 + lane] = nxt[r];
        const double unm = nxt_unm;
        if (rp + 64 < seg_hi) {                // the cells of the next 64 reads: in flight while the group's chains go over this block
#pragma unroll
            for (uint32_t r = 0; r < INIT_TILE_R; r++) if (r < R) nxt[r] = table[row_at[r] + rp + 64];
            nxt_unm = V.table_unm[rp + 64];
        }
        // per read, shared by the chains: the multiples of the golden ratio inside counter_u64(seed, rp << 16 | t) and counter_u64(seed', rp)
        const uint64_t zbase = ((static_cast<uint64_t>(rp) << 16) + 1ull) * GOLDEN64;
        const uint64_t abase = (static_cast<uint64_t>(rp) + 1ull) * GOLDEN64;
        for (uint32_t c = 0; c < T; c++) {
            const InitChainP P = cp[c];
            const uint4 qa = stage[P.ia * 64 + lane], qb = stage[P.ib * 64 + lane];
            const double aln_before = aln[c * 256 + tid];
            const double lpA = __hiloint2double(static_cast<int>(qa.y), static_cast<int>(qa.x));
            const double lpB = __hiloint2double(static_cast<int>(qb.y), static_cast<int>(qb.x));
            const uint32_t rawA = qa.z >> 24, rawB = qb.z >> 24;
            const double thresh = fmax(fmax(lpA, lpB), unm) - V.prob_diff;           // max(unm - d, best_i - d, ...) == max(...) - d
            const bool kA = in && rawA != 0 && lpA >= thresh, kB = in && rawB != 0 && lpB >= thresh, kU = in && unm >= thresh;
            const bool multi = (kA && rawA > 1) || (kB && rawB > 1);                 // several pair-alignments on a contig in reach: deferred
            const uint32_t nw = (kA ? 1u : 0u) + (kB ? 1u : 0u) + (kU ? 1u : 0u);
            bool nontrivial = nw > 1;
            if (__any(multi && nw == 1)) {
                // the contig's second pair-alignment decides whether the read is non-trivial (locs_from_cells: pa[ext + k - 1] at k = 1)
                if (multi && nw == 1) {
                    const uint32_t ext = V.table_ext[static_cast<uint64_t>(kA ? P.row0 : P.row1) * V.ngp + rp];
                    nontrivial = V.pa[ext].ln_prob >= thresh;
                }
            }
            // ordered compaction of the non-trivial reads of this range (assgn.rs:61-63)
            const unsigned long long nt_mask = __ballot(nontrivial);
            const uint32_t before = __builtin_amdgcn_readlane(n_recs_all, c);
            const uint32_t slot = before + static_cast<uint32_t>(__popcll(nt_mask & ((1ull << lane) - 1ull)));
            {
                const uint32_t after = __builtin_amdgcn_readfirstlane(before + static_cast<uint32_t>(__popcll(nt_mask)));
                uint32_t m0_was;                                                   // (one scalar register per VOP3 on gfx9: the lane number goes through m0, which is put back)
                asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
                             : "+v"(n_recs_all), "=&s"(m0_was) : "s"(after), "s"(c));
            }
            const unsigned long long mm = __ballot(multi);
            if (mm) {
                if (multi)
                    __hip_atomic_store(&queue[(qhead + qn + static_cast<uint32_t>(__popcll(mm & ((1ull << lane) - 1ull)))) & (INIT_TILE_Q - 1)],
                                       static_cast<unsigned long long>(rp | (c << 24)) | (static_cast<unsigned long long>(slot) << 32), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WAVEFRONT);
                qn += static_cast<uint32_t>(__popcll(mm));
            }
            // the read's locations in decreasing ln_prob, ties in push order (contig 0, contig 1, then "unmapped") — windows.rs:793
            const bool BgtA = lpB > lpA, UgtA = unm > lpA, UgtB = unm > lpB;
            const uint32_t posA = ((kB && BgtA) ? 1u : 0u) + ((kU && UgtA) ? 1u : 0u);
            const uint32_t posB = ((kA && !BgtA) ? 1u : 0u) + ((kU && UgtB) ? 1u : 0u);
            // get_shifted_window_ix + middle_window (windows.rs:62-68, 465-470) with define_windows_random (123-136)
            uint32_t tA1 = 0, tA2 = 0, tB1 = 0, tB2 = 0;
            if (V.tweak) {
                const uint64_t z0 = P.seed + zbase;
                const uint64_t rA = counter_fin(z0 + (posA == 0 ? 0ull : posA == 1 ? GOLDEN64 : 2ull * GOLDEN64));
                const uint64_t rB = counter_fin(z0 + (posB == 0 ? 0ull : posB == 1 ? GOLDEN64 : 2ull * GOLDEN64));
                tA1 = V.by_tweak.mod(static_cast<uint32_t>(rA >> 32)); tA2 = V.by_tweak.mod(static_cast<uint32_t>(rA));
                tB1 = V.by_tweak.mod(static_cast<uint32_t>(rB >> 32)); tB2 = V.by_tweak.mod(static_cast<uint32_t>(rB));
            }
            auto window_ix = [&](uint32_t mid, bool none, uint32_t tw, uint32_t rs, uint32_t re, uint32_t sh) -> uint32_t {
                const uint32_t m = mid + tw;
                uint32_t q = V.by_window.div(m - rs) + sh;
                asm volatile("" : "+v"(q));                                         // a value, then selects: no branch around the division
                const uint32_t inside = (rs <= m && m < re) ? q : 1u;              // BOUNDARY_WINDOW
                return none ? 0u : inside;                                          // UNMAPPED_WINDOW
            };
            const uint32_t midA1 = qa.z & MID_NONE24, midB1 = qb.z & MID_NONE24;
            const uint32_t winA = window_ix(midA1, midA1 == MID_NONE24, tA1, P.rs0, P.re0, P.shift0) |
                                  (window_ix(qa.w, qa.w == NONE32S, tA2, P.rs0, P.re0, P.shift0) << 16);
            const uint32_t winB = window_ix(midB1, midB1 == MID_NONE24, tB1, P.rs1, P.re1, P.shift1) |
                                  (window_ix(qb.w, qb.w == NONE32S, tB2, P.rs1, P.re1, P.shift1) << 16);
            // the location at place q of the order (q < nw): contig 0, contig 1, or — neither — "both unmapped" (values, selects)
            const bool a_0 = kA && posA == 0, b_0 = kB && posB == 0, a_1 = kA && posA == 1, b_1 = kB && posB == 1, a_2 = kA && posA == 2, b_2 = kB && posB == 2;
            const double lp0 = a_0 ? lpA : (b_0 ? lpB : unm), lp1 = a_1 ? lpA : (b_1 ? lpB : unm), lp2 = a_2 ? lpA : (b_2 ? lpB : unm);
            const uint32_t win0 = a_0 ? winA : (b_0 ? winB : 0u), win1 = a_1 ? winA : (b_1 ? winB : 0u), win2 = a_2 ? winA : (b_2 ? winB : 0u);
            uint32_t a0 = 0;
            double lps = lp0; uint32_t wins = win0;                                // where the read starts: its best location, or a drawn one
            if (random_start) {
                const uint64_t r = counter_fin((P.seed ^ INIT_KEY_XOR) + abase);
                a0 = nw > 1 ? static_cast<uint32_t>(__umul64hi(r, static_cast<uint64_t>(nw))) : 0u;
                lps = a0 == 0 ? lp0 : (a0 == 1 ? lp1 : lp2); wins = a0 == 0 ? win0 : (a0 == 1 ? win1 : win2);
            }
            const bool fast = !multi && nw != 0;
            if (fast) {
                atomicAdd(&depth[c * W + (wins & 0xFFFFu)], 1u);
                atomicAdd(&depth[c * W + (wins >> 16)], 1u);
                aln[c * 256 + tid] = aln_before + lps;
            }
            uint32_t eix = 0;
            if (fast && nw == 3) {
                eix = atomicAdd(&ex_cnt[c], 1u);
                if (eix < V.extra_cap) {
                    ExtraLoc e; e.lp = lp2; e.win = win2; e._pad = 0;
                    V.extra[static_cast<uint64_t>(P.chain) * V.extra_cap + eix] = e;
                }
            }
            if (fast && nw > 1) {
                uint4* dst = reinterpret_cast<uint4*>(V.recs + static_cast<uint64_t>(P.chain) * V.rstride + static_cast<uint64_t>(wave) * V.seg_reads + slot);
                // the record as its two 16-byte halves (left to itself the compiler stores 8 + 16 + 8 bytes, the middle piece unaligned)
                store_16(dst, make_uint4(rp | (a0 << 24), nw | (eix << 8), static_cast<uint32_t>(__double2loint(lp0)), static_cast<uint32_t>(__double2hiint(lp0))));
                store_16(dst + 1, make_uint4(static_cast<uint32_t>(__double2loint(lp1)), static_cast<uint32_t>(__double2hiint(lp1)), win0, win1));
            }
        }
        // The list of deferred reads is worked off between blocks, 64 entries at a time, all of it behind the wavefront's last block: a
        // block adds at most 64 entries per chain to the fewer than 64 left over. (Inside the chain loop the general code's loads made
        // every chain wait for the record stores of the chain before it: the loop holds stores only.)
        const bool last = base + 64 >= seg_hi;
        if (qn >= 64 || (last && qn)) {
            // the list's entries were stored by other lanes of this wavefront: workgroup scope — the stores have completed before the
            // entries are read (same CU, same L1). NOT agent scope: on this multi-XCD device an agent-scope release writes the XCD's L2
            // back (buffer_wbl2) and the acquire invalidates it — with the records streaming through, 57 -> 205 ms for the launch.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            while (qn >= 64 || (last && qn)) {
                const uint32_t n = min(qn, 64u);
                init_tile_deferred(V, cp, T, depth, aln_def, ex_cnt, queue, qhead, n, wave, lane, random_start);
                qhead = (qhead + n) & (INIT_TILE_Q - 1); qn -= n;
            }
        }
    }
    if (lane < T) seg_cnt[lane * 4 + wave] = n_recs_all;
    __syncthreads();
    for (uint32_t c = 0; c < T; c++) {
        const InitChainP P = cp[c];
        const uint32_t ex_total = ex_cnt[c];
        if (tid == 0) {
            if (ex_total > V.extra_cap) { atomicMax(V.overflow, 4u); atomicMax(V.overflow + 1, ex_total); }
            if (ex_total >= (1u << 24)) atomicMax(V.overflow, 2u);
        }
        uint32_t* gdepth = V.c_depth + static_cast<uint64_t>(P.chain) * W;
        for (uint32_t w = tid; w < P.total_w; w += 256) gdepth[w] = depth[c * W + w];
        // c_aln: the threads' shares, added up by the first wavefront
        if (wave == 0) {
            double v = (aln[c * 256 + lane] + aln[c * 256 + 64 + lane]) + (aln[c * 256 + 128 + lane] + aln[c * 256 + 192 + lane]);
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            v += (aln_def[c * 4] + aln_def[c * 4 + 1]) + (aln_def[c * 4 + 2] + aln_def[c * 4 + 3]);
            if (lane == 0) {
                uint32_t* cum = V.c_seg + static_cast<uint64_t>(P.chain) * 4;
                uint32_t run = 0;
                for (uint32_t k = 0; k < INIT_SEGS; k++) { cum[k] = run; run += seg_cnt[c * 4 + k]; }
                V.c_aln[P.chain] = v; V.c_nnt[P.chain] = run; V.c_totw[P.chain] = P.total_w;
            }
        }
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

// An entry of the depth table by its 32-bit BYTE offset: the load then takes the table's address from scalar registers and the offset
// from one vector register (as an index it took a 64-bit shift-and-add per gather: ~40 of the greedy iteration's vector instructions).
// ensure_depth_table keeps the table below 4 GB.
__device__ __forceinline__ double lut_at(const double* lut, uint32_t index) {
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(lut) + (index << 3));
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
        return weight * lut_at(V->lut, g * V->lut_depth + d);
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
            g.vnew[i] = lut_at(V->lut, row + min(d_new, last));
            g.vold[i] = lut_at(V->lut, row + min(d_old, last));
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
            g.vnew[i] = lut_at(V->lut, row + min(d_new, last));
            g.vold[i] = lut_at(V->lut, row + min(d_old, last));
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
        return weight * lut_at(V->lut, (half & 0x7Fu) * V->lut_depth + d);
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
            g.vnew[i] = lut_at(V->lut, row + min(d_new, last));
            g.vold[i] = lut_at(V->lut, row + min(d_old, last));
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
            g.vnew[i] = lut_at(V->lut, row + min(d_new, last));
            g.vold[i] = lut_at(V->lut, row + min(d_old, last));
            g.dmax[i] = max(d_new, d_old);
        }
#pragma unroll
        for (int i = 0; i < 2; i++) g.weight[i] = weight_of(word[i], half[i]);
    }
};

// the `cur` word of a record. Wavefront scope: a chain's records are written by the one wavefront that runs the chain and read back by that
// wavefront (a wavefront sees its own stores: the CU's L1 is written through and kept current); other kernels see them at the kernel's end.
// (At agent scope, as in rounds 1-3, every look at the word was a request of its own to the L2 next to the record's line: 50 of the greedy
// loop's 304 L1 -> L2 requests per wavefront-iteration.)
__device__ __forceinline__ uint32_t load_rp_cur(const ChainRec* r) {
    return __hip_atomic_load(&r->rp_cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ __forceinline__ void store_rp_cur(ChainRec* r, uint32_t v) {
    __hip_atomic_store(&r->rp_cur, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
// the immutable part of a record: two 16-byte loads
struct RecBody { uint32_t meta; double lp0, lp1; uint32_t win0, win1; };
__device__ __forceinline__ RecBody load_body(const ChainRec* r, uint32_t* rp_cur = nullptr) {
    const uint4 a = *reinterpret_cast<const uint4*>(r);
    const uint4 b = *(reinterpret_cast<const uint4*>(r) + 1);
    if (rp_cur) *rp_cur = a.x;                                              // the word in front of the body: no load of its own
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
    uint32_t packed;
    const RecBody b = load_body(&recs[m.slot], &packed);                    // (the chain's own stores are in it: wavefront scope, see load_rp_cur)
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

constexpr uint32_t GREEDY_ROW_TAIL = 8;       // words per row behind the windows (see the kernel)
// LDS of a greedy workgroup for its rows' windows: 4 bytes each; 6 with the weights in LDS, plus the two weight tables (16-byte multiple)
__host__ __device__ inline size_t greedy_lds_windows(uint32_t lpc, uint32_t wstride, uint32_t n_wk, uint32_t n_wc, bool lw) {
    const size_t rows = static_cast<size_t>(64 / lpc) * (lw ? 2 : 1) * wstride;
    const size_t b = lw ? ((rows * 6 + 7) & ~static_cast<size_t>(7)) + static_cast<size_t>(n_wk + n_wc) * 8 : rows * 4;
    return (b + 15) & ~static_cast<size_t>(15);
}

// ---------------- K14 Greedy: 64 / LPC chains per wavefront ----------------
// A candidate read of an iteration as its lane sees it: the record, and the first two of its locations beyond the second
// as they were loaded: the two 16-byte halves of a ChainRec, two ExtraLoc entries. They are taken apart where an iteration uses them (a
// 16-byte load taken apart right behind itself makes the compiler wait for it there)
struct GreedyCand { uint32_t pick; uint4 q0, q1; };
struct GreedyExtRaw { uint4 x0, x1; };
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
    // per row, behind the windows: [0..3] the parts of the record list, [4..5] row_best
    uint32_t* row_cum = reinterpret_cast<uint32_t*>(smem + greedy_lds_windows(LPC, V.wstride, V.n_wk, V.n_wc, LW)) + static_cast<size_t>(wg_row) * GREEDY_ROW_TAIL;
    if (jj < 4) row_cum[jj] = V.c_seg[static_cast<uint64_t>(chain) * 4 + jj];
    [[maybe_unused]] double* row_best = reinterpret_cast<double*>(row_cum + 4);          // [4..5] the row's largest improvement of an iteration
    if (jj == 0) *row_best = -INFINITY;
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
                // two lanes at ring distance k meet at rotation k or LPC - k, one of which is at most LPC / 2 (all LPC - 1 rotations: + 4 ms)
                for (uint32_t d = 1; d <= LPC / 2; d++) {
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
        // A record is requested as its two 16-byte halves, its further locations as two 16-byte entries: as six + four field loads (rounds
        // 2-3) they were ten look-ups in the CU's L1 per candidate for two lines, and the current-location word, read at agent scope, a request
        // of its own to the L2. Round 4: 957 -> 671 L1 look-ups and 304 -> 247 L1 -> L2 requests per wavefront-iteration, 303 -> 276 ms for the
        // stage alone. (What the loop waits for is the CU's L1 path, not its own instructions: profiles/r04_greedy_forms_and_phases.txt.)
        auto request_record = [&](GreedyCand& c) {
            c.pick = sample();
            const uint4* r = reinterpret_cast<const uint4*>(&recs[cand ? c.pick : 0u]);
            c.q0 = r[0]; c.q1 = r[1];
        };
        // the first two further locations of a record that has arrived (reads with two locations: the chain's first entry, unused)
        auto request_ext = [&](const GreedyCand& c, GreedyExtRaw& e) {
            const uint32_t meta = c.q0.y, nloc = meta & 0xFFu;
            const uint4* p = reinterpret_cast<const uint4*>(extra + (cand && nloc > 2 ? (meta >> 8) : 0u));   // the run has spare entries behind it
            // (unconditional: under `if (__any(nloc > 2))` the loop took 306 ms instead of 274, under a per-lane `if` 292 — the wait counts
            // in front of the gathers' use must then fit both paths; profiles/r06_solver_notes.txt section 5)
            e.x0 = p[0]; e.x1 = p[1];
        };
        uint32_t curr_plato = 0;
        uint64_t iter = 0;
        // the last three moves of the row, newest first (slot 0xFFFFFFFF: none)
        uint32_t h1s = 0xFFFFFFFFu, h1t = 0, h2s = 0xFFFFFFFFu, h2t = 0, h3s = 0xFFFFFFFFu, h3t = 0;

        // one iteration: A = its candidates (arrived), EA their further locations (arrived); N = the candidates of the next
        // iteration (arrived), EN receives their further locations; F = the slot the sample three iterations ahead goes to (= A's)
        auto iteration = [&](auto slot_tag, GreedyCand& A, const GreedyExtRaw& EA, const GreedyCand& N, GreedyExtRaw& EN) {
            [[maybe_unused]] constexpr uint32_t SLOT = decltype(slot_tag)::value;
            constexpr bool TIMED = (GREEDY_FORM & 32u) != 0;
            uint64_t tk[6] = {0, 0, 0, 0, 0, 0}, ts[3] = {0, 0, 0}, tw0 = 0;
            auto subtick = [&](int k) { if constexpr (TIMED) { __builtin_amdgcn_sched_barrier(0); ts[k] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } };
            auto tick = [&](int k) { if constexpr (TIMED) { __builtin_amdgcn_sched_barrier(0); tk[k] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } };
            tick(0);
            if constexpr (TIMED) {                     // how long the further locations requested an iteration ago are still on their way
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0); tw0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
            }
            RecBody b; GreedyExt EAu;
            const uint32_t pick = A.pick, rpc0 = A.q0.x;
            {
                const uint4 q0 = A.q0, q1 = A.q1, x0 = EA.x0, x1 = EA.x1;
                b.meta = q0.y; b.lp0 = __hiloint2double(static_cast<int>(q0.w), static_cast<int>(q0.z));
                b.lp1 = __hiloint2double(static_cast<int>(q1.y), static_cast<int>(q1.x)); b.win0 = q1.z; b.win1 = q1.w;
                EAu.lp2 = __hiloint2double(static_cast<int>(x0.y), static_cast<int>(x0.x)); EAu.win2 = x0.z;
                EAu.lp3 = __hiloint2double(static_cast<int>(x1.y), static_cast<int>(x1.x)); EAu.win3 = x1.z;
            }
            const uint32_t nloc = b.meta & 0xFFu;
            uint32_t cur = rpc0 >> 24;
            cur = pick == h3s ? h3t : cur; cur = pick == h2s ? h2t : cur; cur = pick == h1s ? h1t : cur;
            // lanes without a candidate (beyond the sample, rows that are done or spare) look at window 0 four times: no change, no effect
            const uint32_t n_alt = (cand && !done) ? nloc - 1 : 0u;
            const bool deep = n_alt != 0 && nloc > GREEDY_INLINE_LOCS;           // some locations of this read are not in registers
            double cur_lp = 0.0; uint32_t cur_w = 0;
            if (n_alt) cand_loc(b, EAu, min(cur, GREEDY_INLINE_LOCS - 1), &cur_lp, &cur_w);
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
                    if (has) cand_loc(b, EAu, t_of[u], &lp_t[u], &win_t[u]);
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
            double best;
            if constexpr ((LPC & (LPC - 1)) != 0) {
                // rows of 10 or 12 lanes: the LDS works a wavefront's operations off in order — the row's lanes meet in ONE atomic maximum,
                // the read behind it sees all of them, the reset behind that is seen by the next iteration's (four rounds of two
                // ds_bpermute each on the ring: + 7 ms)
                if (n_alt) __hip_atomic_fetch_max(row_best, my_improv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                best = __hip_atomic_load(row_best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (jj == 0) __hip_atomic_store(row_best, -INFINITY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            } else best = row_max_f64<LPC>(my_improv, row_base, jj);
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
            // release AND acquire: the record loads of the coming iterations (request_record, plain 16-byte loads of other lanes) must not be
            // moved above this store of the current-location word — the three-move history covers the loads issued BEFORE it only
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
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
        GreedyExtRaw E0, E1, E2;
        if constexpr ((GREEDY_FORM & 32u) != 0) t_loop0 = __builtin_amdgcn_s_memtime();
        // in the order the loop leaves its requests behind — the further locations of the coming iteration, THEN the record of the one
        // after the next: the first copy of the unrolled loop is entered from here and from the loop's end, and the wait in front of
        // its further locations must fit both (with the records requested first it waited for everything in flight, one iteration in three)
        request_record(R0); request_record(R1);
        request_ext(R0, E0);
        request_record(R2);
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
    // the annealing wavefronts issue first next to the greedy chains of the following locus (round 3: 537 ms per step; greedy first 567, no
    // priorities 579; round 4, with the annealing stage the shorter of the two: 466.7 ms, greedy first 468.3 with the annealing stage 36 ms longer)
    __builtin_amdgcn_s_setprio(3);
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
                    uint32_t rp;
                    const RecBody b = load_body(&recs[slot], &rp);
                    rp &= 0xFFFFFFu;
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
        // following reads see these updates without waiting for the store to complete (no s_waitcnt vmcnt(0) here); acquire as well:
        // no later load of a record (pf_issue, load_body) may be moved above the store
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
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
        constexpr uint32_t MAXA = 6;                                          // accepted moves one round of the second loop applies at most
        uint32_t pf = 0, pf_slot = 0xFFFFFFFFu, pf_base = 0, pf_n = 0;
        // the moves applied since the `cur` words were requested (uniform): the words were read before those stores and are patched when
        // they are taken — not where the moves are applied: that would wait for the words, which come from HBM
        uint32_t lm_slot[MAXA], lm_to[MAXA];
#pragma unroll
        for (uint32_t k = 0; k < MAXA; k++) { lm_slot[k] = 0xFFFFFFFFu; lm_to[k] = 0; }
        auto pf_issue = [&]() {                                               // one load per lane, no branch around it
            const uint32_t avail = __hip_atomic_load(&ring->produced, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - consumed;
            pf_base = consumed; pf_n = min(avail, 64u);
            pf_slot = lane < pf_n ? slot_at(lane) : 0u;
            pf = load_rp_cur(&recs[pf_slot]);
#pragma unroll
            for (uint32_t k = 0; k < MAXA; k++) lm_slot[k] = 0xFFFFFFFFu;
        };
        auto pf_covered = [&]() -> uint32_t {                                 // positions from `consumed` on whose words are here
            const uint32_t end = pf_base + pf_n;
            return end - consumed <= 64u ? end - consumed : 0u;                // (unsigned: end < consumed wraps to a large number)
        };
        auto pf_take = [&](uint32_t off) -> uint32_t {                        // every lane calls it; valid for off < pf_covered()
            const int src = static_cast<int>((consumed + off - pf_base) & 63u);
            const uint32_t v = static_cast<uint32_t>(__shfl(static_cast<int>(pf), src));
            const uint32_t vs = static_cast<uint32_t>(__shfl(static_cast<int>(pf_slot), src));
            uint32_t out = v;
#pragma unroll
            for (uint32_t k = 0; k < MAXA; k++) out = vs == lm_slot[k] ? (v & 0xFFFFFFu) | (lm_to[k] << 24) : out;
            return out;
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
        // Both loops of stoch.rs:214-241 as ROUNDS over the staged positions of the random stream. A move changes the state only when it
        // is accepted, so the moves that follow a rejection see the same state: lane q evaluates the move that starts at draw q of the
        // stream, the lanes that lie on the true chain of moves are walked in order, and every accepted move is applied up to the first
        // position that meets an earlier accepted move of the round (it starts the next round, evaluated afresh). Same moves, same order,
        // same sums as the serial loops.
        //   second loop (228-241): a move takes one draw, two when its read has more than two locations; accepted iff improvement > min_diff.
        //   first loop (214-226, round 5): the Metropolis test takes one more draw when the move does not pay (diff < 0), so where the
        //     chain goes on depends on the evaluations — known for every position once the lanes have evaluated theirs: a scalar walk over
        //     three ballots; a position's step number i (its temperature) is `steps_left` minus the positions on the chain before it.
        uint32_t steps_left = V.solver.anneal_steps;                       // first loop: the step about to be taken (i of stoch.rs:214)
        bool phase1 = true;
        uint64_t iter = 0;
        uint32_t width = 16;                                               // lanes that speculate: about twice the recent run length
        // (developer build, knob "solve_anneal_timing": SolveView::dbg set) shader-clock ticks of the phases of a round, rounds, moves
        const bool timed = V.dbg != nullptr;
        uint64_t tph[5] = {0, 0, 0, 0, 0}, n_rounds = 0, n_walked = 0, t_first = timed ? __builtin_amdgcn_s_memtime() : 0;
        auto stamp = [&]() -> uint64_t { return timed ? __builtin_amdgcn_s_memtime() : 0ull; };
        while (!lost) {
            if (phase1 && steps_left == 0) phase1 = false;
            if (!phase1 && !(iter < max_iter && curr_plato < V.solver.plato_size)) break;
            const uint64_t ta = stamp();
            const uint32_t need = phase1 ? 3u : 2u;                        // a move may take the draw after its own, the Metropolis test one more
            const uint32_t avail = ring_wait(need);
            if (!avail) break;
            if (pf_covered() < 2) pf_issue();
            // a round may take several moves when neither the loop's count nor the plateau can end inside it (it walks at most 63 positions)
            const bool many_ok = V.solver.plato_size > 64u && curr_plato + 64u < V.solver.plato_size &&
                                 (phase1 ? steps_left > 64u : iter + 64u <= max_iter);
            uint32_t w = min(min(width, avail - (need - 1u)), pf_covered());   // only positions whose `cur` words are here
            if (phase1 && !many_ok) w = min(w, 1u);                         // the schedule's last steps: one position a round
            const bool many = phase1 || many_ok;
            Move m; blank(m);
            bool accepted = false, wide = false;
            const uint64_t tb = stamp();
            const uint32_t packed = pf_take(lane);
            if (lane < w) wide = move_pre(lane, packed, m) == 2;                // lanes beyond: windows 0, no effect
            if (timed) asm volatile("" :: "v"(m.w1), "v"(m.w3), "v"(m.lp_new));
            const uint64_t tc = stamp();
            {
                typename ChainT::DepthGather g;
                C.request(m.w1, m.w2, m.w3, m.w4, g);
                pf_issue();                                                     // the words of the next round travel with this round's gathers
                m.ddiff = C.finish(g);
            }
            const double impr = improvement(m);
            const double diff = impr - min_diff;                               // first loop (stoch.rs:216)
            const bool needu = phase1 && lane < w && !(diff >= 0.0);            // the Metropolis draw is taken (217: `||` evaluates its right side)
            if (!phase1) accepted = lane < w && impr > min_diff;
            if (timed) asm volatile("" :: "v"(m.ddiff));
            const uint64_t td = stamp();
            const unsigned long long two = __ballot(wide);
            const unsigned long long nu = __ballot(needu);
            uint32_t q = 0, walked = 0;
            int hit = -1;
            const unsigned long long below = lane ? ((1ull << lane) - 1ull) : 0ull;
            unsigned long long chain_mask;
            if (phase1) {
                // position p on the chain is followed by p + 1 + (two draws for the move) + (the Metropolis draw)
                const unsigned long long two_u = uniform64(two), nu_u = uniform64(nu);
                const uint32_t w_u = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(w)));
                unsigned long long cm = 0ull;
                for (uint32_t pos = 0; pos < w_u; pos += 1u + static_cast<uint32_t>((two_u >> pos) & 1ull) + static_cast<uint32_t>((nu_u >> pos) & 1ull)) cm |= 1ull << pos;
                chain_mask = cm;
                const bool on = ((chain_mask >> lane) & 1ull) != 0ull;
                accepted = on && !needu;
                if (on && needu) {
                    const uint32_t i_q = steps_left - static_cast<uint32_t>(__popcll(chain_mask & below));       // >= 1: steps_left > 64 or w == 1
                    accepted = ring_f64(lane + (wide ? 2u : 1u)) <= exp(diff / (temp_step * static_cast<double>(i_q)));
                }
            } else {
                // position 0 is on the chain; position p > 0 is unless p - 1 is and its move takes two draws. Behind the nearest position
                // r < p whose move takes one draw (or the start) the chain alternates, so p is on it iff p - (r + 1) is even: one
                // count-leading-zeros per lane instead of a serial walk over the draws.
                const unsigned long long ones = ~two & below;
                const uint32_t after = ones ? 64u - static_cast<uint32_t>(__clzll(static_cast<long long>(ones))) : 0u;      // r + 1
                chain_mask = __ballot(lane < w && ((lane - after) & 1u) == 0u);
            }
            const unsigned long long acc = __ballot(accepted);
            const unsigned long long hits = acc & chain_mask;
            // SEVERAL ACCEPTED MOVES PER ROUND. A move that follows an accepted one sees another state only where the two meet: its
            // evaluation — the depths of its four windows, the current location of its read — is the one the serial loop would make as
            // long as none of its windows and not its read belong to a move accepted before it in this round.
            unsigned long long applied = 0ull;
            if (many) {
                // (everything that steers this loop is the same in all lanes and told to the compiler as such: left as lane values it
                // becomes a divergent loop of nested branches — 3 700 clock ticks a round instead of a few hundred)
                unsigned long long rest = uniform64(hits);
                uint32_t n_app = 0, stop_excl = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(w)));   // positions from stop_excl on belong to the next round
                const double my_dl = m.lp_new - m.lp_old;
                const uint32_t my_w12 = m.w1 | (m.w2 << 16), my_w34 = m.w3 | (m.w4 << 16);
                const uint32_t on_chain = static_cast<uint32_t>((chain_mask >> lane) & 1ull);
                while (rest) {
                    const uint32_t a = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(__ffsll(static_cast<long long>(rest)) - 1));
                    if (a >= stop_excl || n_app == MAXA) break;
                    const int al = static_cast<int>(a);
                    const uint32_t a12 = __builtin_amdgcn_readlane(my_w12, al), a34 = __builtin_amdgcn_readlane(my_w34, al);
                    const uint32_t aslot = __builtin_amdgcn_readlane(m.slot, al), ato = __builtin_amdgcn_readlane(m.new_assgn, al);
                    const uint32_t aw1 = a12 & 0xFFFFu, aw2 = a12 >> 16, aw3 = a34 & 0xFFFFu, aw4 = a34 >> 16;
                    n_acc++; n_app++;
                    depth_lik += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(m.ddiff), al), __builtin_amdgcn_readlane(__double2loint(m.ddiff), al));
                    aln_lik += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(my_dl), al), __builtin_amdgcn_readlane(__double2loint(my_dl), al));
                    applied |= 1ull << a;
#pragma unroll
                    for (uint32_t k = 0; k < MAXA; k++) if (n_app == k + 1) { lm_slot[k] = aslot; lm_to[k] = ato; }
                    // windows 0 and 1 (unmapped, out of the region) carry no distribution: their depths enter no evaluation.
                    // Bit operations, no short cuts: a chain of || is a chain of branches
                    auto meets = [&](uint32_t x) -> uint32_t {
                        return static_cast<uint32_t>(x >= 2u) & (static_cast<uint32_t>(x == aw1) | static_cast<uint32_t>(x == aw2) | static_cast<uint32_t>(x == aw3) | static_cast<uint32_t>(x == aw4));
                    };
                    const uint32_t conflict = static_cast<uint32_t>(lane > a) & on_chain &
                                              (static_cast<uint32_t>(m.slot == aslot) | meets(m.w1) | meets(m.w2) | meets(m.w3) | meets(m.w4));
                    const unsigned long long cm = __ballot(conflict != 0u);
                    if (cm) stop_excl = min(stop_excl, static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(__ffsll(static_cast<long long>(cm)) - 1)));
                    rest &= ~((2ull << a) - 1ull);                              // a <= 62: w <= RING - 1
                }
                if (rest) stop_excl = min(stop_excl, static_cast<uint32_t>(__ffsll(static_cast<long long>(rest))) - 1u);   // an accepted move this round does not take
                const unsigned long long upto = stop_excl >= 64u ? ~0ull : ((1ull << stop_excl) - 1ull);
                const unsigned long long P = chain_mask & upto;                // the positions the serial loop walks on this state
                const uint32_t moves = static_cast<uint32_t>(__popcll(P));
                n_iter += moves; walked = moves;
                if (phase1) steps_left -= moves; else iter += moves;
                if (applied) {
                    const uint32_t last_a = 63u - static_cast<uint32_t>(__clzll(static_cast<long long>(applied)));
                    curr_plato = static_cast<uint32_t>(__popcll(P & ~((2ull << last_a) - 1ull)));
                } else curr_plato += moves;
                const uint32_t lastp = 63u - static_cast<uint32_t>(__clzll(static_cast<long long>(P)));
                q = lastp + 1u + static_cast<uint32_t>((two >> lastp) & 1ull) + static_cast<uint32_t>((nu >> lastp) & 1ull);
                if ((applied >> lane) & 1ull) {
                    atomicAdd(&wd[m.w3], 1u); atomicAdd(&wd[m.w4], 1u);        // the depth field never borrows from the GC bits
                    atomicSub(&wd[m.w1], 1u); atomicSub(&wd[m.w2], 1u);
                    store_rp_cur(&recs[m.slot], m.rp | (m.new_assgn << 24));
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // stoch.rs:222-224: the schedule ends at a rejection that fills the plateau (only a round of one position can get here)
                if (phase1 && curr_plato >= V.solver.plato_size) phase1 = false;
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
            const uint64_t te = stamp();
            if (hit >= 0) {
                Move a;
                a.rp = __shfl(m.rp, hit); a.new_assgn = __shfl(m.new_assgn, hit); a.slot = __shfl(m.slot, hit);
                a.w1 = __shfl(m.w1, hit); a.w2 = __shfl(m.w2, hit); a.w3 = __shfl(m.w3, hit); a.w4 = __shfl(m.w4, hit);
                a.lp_old = __shfl(m.lp_old, hit); a.lp_new = __shfl(m.lp_new, hit); a.ddiff = __shfl(m.ddiff, hit);
                reassign(a);
                curr_plato = 0;
                lm_slot[0] = a.slot; lm_to[0] = a.new_assgn;                    // the words requested above were read before this store
            }
            if (many) hit = applied ? 0 : -1;                                   // for the width below: something was accepted
            width = min(RING - 1, max(8u, hit >= 0 ? (width + 2 * walked + 4) / 2 : 2 * width));
            retire(q);
            if (timed) {
                const uint64_t tf = stamp();
                tph[0] += tb - ta; tph[1] += tc - tb; tph[2] += td - tc; tph[3] += te - td; tph[4] += tf - te;
                n_rounds++; n_walked += walked;
            }
        }
        if (timed && lane == 0) {
            double* d = V.dbg + static_cast<size_t>(chain) * 12;
            for (int k = 0; k < 5; k++) d[k] = static_cast<double>(tph[k]);
            d[5] = static_cast<double>(__builtin_amdgcn_s_memtime() - t_first); d[6] = static_cast<double>(n_rounds); d[7] = static_cast<double>(n_walked);
            d[8] = static_cast<double>(n_acc); d[9] = static_cast<double>(n_iter);
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

// location table + compact "unmapped" column of a scored batch (rebuilt after every lcty_score_reads)
void ensure_solver_tables(lcty_reads* reads) {
    reads->ensure_good_index();
    // a streaming batch that goes on to the solver is past its chunks: the spare record tables of alignment recovery (lcty_objects.hpp) are
    // device memory the stage's workspace wants (a later recovery allocates them again)
    reads->spare_aln_off.release(); reads->spare_cigar_off.release(); reads->spare_recs.release(); reads->spare_cigar.release();
    reads->spare_pair_meta.release();
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
    if (depth > (1u << 22)) fail(LCTY_ERR_UNSUPPORTED, "a window more than 4 M read ends deep (the depth table is addressed by 32-bit byte offsets)");
    lcty_ctx* ctx = loc->ctx;
    loc->d_lut_ext.alloc(static_cast<size_t>(LCTY_GC_BINS) * depth);
    const uint32_t n = LCTY_GC_BINS * depth;
    hipLaunchKernelGGL(build_depth_table_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, loc->d_depth_lut.p, loc->d_depth_nb.p,
                       static_cast<uint32_t>(loc->prm.n_alt_cn), depth, loc->d_lut_ext.p);
    LCTY_HIP(hipGetLastError());
    loc->lut_ext_depth = depth;
}

// the same table of `loc` into a buffer of the caller's, on the caller's stream: lcty_solve_given keeps one per slot,
// so that calls from several threads never share a table that one of them is widening
void build_depth_table_into(const lcty_locus* loc, uint32_t depth, DevBuf<double>& out, hipStream_t s) {
    if (depth > (1u << 22)) fail(LCTY_ERR_UNSUPPORTED, "a window more than 4 M read ends deep (the depth table is addressed by 32-bit byte offsets)");
    out.alloc(static_cast<size_t>(LCTY_GC_BINS) * depth);
    const uint32_t n = LCTY_GC_BINS * depth;
    hipLaunchKernelGGL(build_depth_table_kernel, dim3((n + 255) / 256), dim3(256), 0, s, loc->d_depth_lut.p, loc->d_depth_nb.p,
                       static_cast<uint32_t>(loc->prm.n_alt_cn), depth, out.p);
    LCTY_HIP(hipGetLastError());
}

// a few microseconds of nothing (one wavefront): lets the workgroups of a kernel launched just before on another stream get resident
__global__ void pause_kernel(uint32_t rounds) {
    for (uint32_t i = 0; i < rounds; i++) __builtin_amdgcn_s_sleep(127);
}

template <uint32_t P>
void launch_init_p(lcty_ctx* ctx, const SolveView& V, uint32_t nch, size_t lds_init, hipStream_t s) {
    if (lds_init > 48 * 1024)
        LCTY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(solve_init_kernel<P>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(lds_init)));
    ctx->timed(V.solver.kind == LCTY_SOLVER_ANNEAL ? LCTY_K_SOLVE_INIT_ANNEAL : LCTY_K_SOLVE_INIT,
               [&] { hipLaunchKernelGGL(solve_init_kernel<P>, dim3(nch), dim3(256), lds_init, s, V); }, s);
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
#ifdef LCTY_DIAG
    if constexpr (LPC == 12 && LW) {
        // the developer build: the loop with shader-clock stamps between its phases (knob "solve_greedy_form" 32), for the stage shape of
        // the default scheme only
        switch (ctx->diag_knob("solve_greedy_form", 0)) {
            case 32: return launch_greedy_form<LPC, LW, 32>(ctx, V, nch, s);
            default: break;
        }
    }
#endif
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

// Which form of the greedy loop a stage takes (lanes per chain, window weights from LDS tables) and its launch.
void launch_greedy_chains(lcty_ctx* ctx, SolveView& V, uint32_t nch, hipStream_t stream, lcty_ctx::SolveWorkspace& ws) {
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
    const bool timed_form = lpc == 12 && lw && (ctx->diag_knob("solve_greedy_form", 0) & 32) != 0;
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

// ---------------------------------------------------------------- launchers (declared in lcty_solve_device.hpp)
bool solver_lds_fits(uint32_t wstride) {
    return !(static_cast<size_t>(wstride) * 12 + sizeof(AnnealRing) + 128 > 160 * 1024 || wstride > 65535);
}
// The groups of a diploid stage (solve_init_tile_kernel). Alleles are ranked by how many of the stage's genotypes hold them; a genotype
// whose better-ranked allele has rank u and whose other allele has rank v belongs to tile (u / 2, v / 4): two alleles by four, eight
// genotypes on six rows when the tile is full — the top of a prefilter is a few alleles paired with nearly everything plus a dense
// core, and both fill their tiles. Genotypes are taken in tile order and a group is closed when the next genotype's chains or rows
// no longer fit; the attempts of a genotype stay together (they share both rows) unless there are more of them than a group holds.
//
// How long the launch lasts is a matter of how its workgroups fill the device, not only of what they read: a workgroup lasts as
// long as its chains are many (the kernel is bound by its instruction stream), S of them are resident, and 1 091 groups of five on
// 512 places are three rounds where 1 024 would have been two. So: the groups are as large as "every place gets one" allows
// (ceil(chains / S), at most INIT_TILE_T), the largest go first, and what is left behind the first S groups is cut into pieces small
// enough that every place gets three or more of them — the places that finish their large group take small ones until none is left,
// and the launch ends within one small piece of the ideal.
void plan_init_groups(const SolveView& V, const InitHost& H, uint32_t nch, InitPlan& plan) {
    plan.groups.clear(); plan.chains.clear(); plan.T = plan.R = 0; plan.lds = 0;
    const uint32_t attempts = V.attempts, ng = (nch + attempts - 1) / attempts, A = V.A;
    uint32_t T_cap = INIT_TILE_T;                                            // what the LDS holds with two workgroups per CU
    const size_t lds_budget = H.lds_budget ? H.lds_budget : 80 * 1024;
    while (T_cap > 1 && init_tile_lds(T_cap, std::min(INIT_TILE_R, T_cap + 1), V.wstride) > lds_budget) T_cap--;
    const uint64_t per_cu = std::max<uint64_t>(1, std::min<uint64_t>(160 * 1024 / init_tile_lds(T_cap, std::min(INIT_TILE_R, T_cap + 1), V.wstride), 4));
    const uint64_t places = per_cu * std::max(1u, H.n_cus);
    T_cap = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(T_cap, (nch + places - 1) / places)));
    const uint32_t R_cap = std::min(INIT_TILE_R, T_cap + 1);
    std::vector<uint32_t> deg(A, 0), rank(A, 0), order(A);
    for (uint32_t g = 0; g < ng; g++) { deg[H.genotypes[2 * g]]++; deg[H.genotypes[2 * g + 1]]++; }
    for (uint32_t a = 0; a < A; a++) order[a] = a;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return deg[x] > deg[y]; });
    for (uint32_t r = 0; r < A; r++) rank[order[r]] = r;
    std::vector<uint32_t> gts(ng);
    std::vector<uint64_t> key(ng);
    for (uint32_t g = 0; g < ng; g++) {
        const uint32_t ra = rank[H.genotypes[2 * g]], rb = rank[H.genotypes[2 * g + 1]];
        const uint64_t u = std::min(ra, rb), v = std::max(ra, rb);
        key[g] = ((u / 2) << 48) | ((v / 4) << 32) | (u << 16) | v;
        gts[g] = g;
    }
    std::sort(gts.begin(), gts.end(), [&](uint32_t x, uint32_t y) { return key[x] != key[y] ? key[x] < key[y] : x < y; });
    auto row_of = [&](uint32_t allele) -> uint32_t { return H.row_of ? H.row_of[allele] : allele; };
    // first the groups as lists of (genotype, first attempt, attempts) ...
    struct Unit { uint32_t gi, a0, n; };
    struct Proto { std::vector<Unit> units; uint32_t n_chains = 0, n_rows = 0; uint32_t row[8] = {0, 0, 0, 0, 0, 0, 0, 0}; };
    std::vector<Proto> protos;
    Proto cur;
    auto close = [&] { if (cur.n_chains) protos.push_back(cur); cur = Proto{}; };
    auto fresh_rows = [&](const Proto& p, uint32_t r0, uint32_t r1) -> uint32_t {
        bool h0 = false, h1 = false;
        for (uint32_t r = 0; r < p.n_rows; r++) { h0 |= p.row[r] == r0; h1 |= p.row[r] == r1; }
        return (h0 ? 0u : 1u) + ((h1 || r1 == r0) ? 0u : 1u);
    };
    auto add_rows = [&](Proto& p, uint32_t r0, uint32_t r1) {
        for (uint32_t x : {r0, r1}) {
            bool have = false;
            for (uint32_t r = 0; r < p.n_rows; r++) have |= p.row[r] == x;
            if (!have) p.row[p.n_rows++] = x;
        }
    };
    for (uint32_t gi : gts) {
        const uint32_t r0 = row_of(H.genotypes[2 * gi]), r1 = row_of(H.genotypes[2 * gi + 1]);
        for (uint32_t a0 = 0; a0 < attempts;) {
            const uint32_t want = std::min(attempts - a0, T_cap);
            if (cur.n_chains + want > T_cap || cur.n_rows + fresh_rows(cur, r0, r1) > R_cap) close();
            add_rows(cur, r0, r1);
            cur.units.push_back(Unit{gi, a0, want}); cur.n_chains += want;
            a0 += want;
        }
    }
    close();
    // ... the largest first, the tail behind the first `places` of them in small pieces ...
    std::stable_sort(protos.begin(), protos.end(), [](const Proto& a, const Proto& b) { return a.n_chains > b.n_chains; });
    if (protos.size() > places) {
        uint64_t tail_chains = 0;
        for (size_t i = places; i < protos.size(); i++) tail_chains += protos[i].n_chains;
        const uint32_t piece = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(T_cap, tail_chains / (3 * places))));
        std::vector<Proto> tail;
        for (size_t i = places; i < protos.size(); i++) {
            Proto part;
            for (const Unit& u : protos[i].units)
                for (uint32_t done = 0; done < u.n;) {
                    const uint32_t take = std::min(u.n - done, piece - part.n_chains);
                    add_rows(part, row_of(H.genotypes[2 * u.gi]), row_of(H.genotypes[2 * u.gi + 1]));
                    part.units.push_back(Unit{u.gi, u.a0 + done, take}); part.n_chains += take;
                    done += take;
                    if (part.n_chains == piece) { tail.push_back(part); part = Proto{}; }
                }
            if (part.n_chains) tail.push_back(part);
        }
        protos.resize(places);
        protos.insert(protos.end(), tail.begin(), tail.end());
    }
    // ... then the groups and their chains as the kernel reads them
    for (const Proto& p : protos) {
        InitGroup g{};
        g.first = static_cast<uint32_t>(plan.chains.size()); g.n_rows = p.n_rows;
        for (uint32_t r = 0; r < 8; r++) g.row[r] = p.row[r];
        for (const Unit& u : p.units) {
            const uint32_t id0 = H.genotypes[2 * u.gi], id1 = H.genotypes[2 * u.gi + 1];
            const uint32_t row0 = row_of(id0), row1 = row_of(id1);
            uint32_t ia = 0, ib = 0;
            for (uint32_t r = 0; r < p.n_rows; r++) { if (p.row[r] == row0) ia = r; if (p.row[r] == row1) ib = r; }
            const uint32_t nw0 = H.loc->n_windows[id0], nw1 = H.loc->n_windows[id1];
            for (uint32_t k = 0; k < u.n; k++) {
                const uint32_t chain = u.gi * attempts + u.a0 + k;
                if (chain >= nch) break;
                InitChainP P{};
                P.seed = H.seeds[chain]; P.chain = chain; P.ia = ia; P.ib = ib; P.row0 = row0; P.row1 = row1; P.id0 = id0; P.id1 = id1;
                P.shift0 = 2; P.rs0 = H.loc->reg_start[id0]; P.re0 = P.rs0 + nw0 * V.window;           // GenotypeWindows: REG_WINDOW_SHIFT = 2
                P.shift1 = 2 + nw0; P.rs1 = H.loc->reg_start[id1]; P.re1 = P.rs1 + nw1 * V.window;
                P.total_w = 2 + nw0 + nw1;
                plan.chains.push_back(P);
                g.n_chains++;
            }
        }
        if (!g.n_chains) continue;
        plan.T = std::max(plan.T, g.n_chains); plan.R = std::max(plan.R, g.n_rows);
        plan.groups.push_back(g);
    }
    plan.lds = init_tile_lds(plan.T, plan.R, V.wstride);
}

void launch_init(lcty_ctx* ctx, const SolveView& V, uint32_t nch, size_t lds_init, hipStream_t s, const InitHost* host, lcty_ctx::SolveWorkspace& ws,
                 InitPlan& plan) {
    if (V.ploidy == 2 && host && nch && ctx->diag_knob("solve_init_tiles", 1) != 0) {
        plan_init_groups(V, *host, nch, plan);
        const size_t gb = plan.groups.size() * sizeof(InitGroup), cb = plan.chains.size() * sizeof(InitChainP);
        if (ctx->diag_knob("solve_stats", 0)) {
            uint64_t rows = 0; uint32_t hist[INIT_TILE_T + 1] = {0};
            for (const InitGroup& g : plan.groups) { rows += g.n_rows; hist[g.n_chains]++; }
            fprintf(stderr, "[lcty solve] initialisation: %u chains in %zu groups (largest %u chains, %u rows; %.2f rows per chain; LDS %zu B); groups of 1..%u chains:",
                    nch, plan.groups.size(), plan.T, plan.R, static_cast<double>(rows) / nch, plan.lds, INIT_TILE_T);
            for (uint32_t t = 1; t <= INIT_TILE_T; t++) fprintf(stderr, " %u", hist[t]);
            fprintf(stderr, "\n");
        }
        const size_t lb = plan.groups.size() * 4 * INIT_TILE_Q * sizeof(unsigned long long);   // a list of deferred reads per wavefront
        ws.init_plan.ensure_slack(gb + cb + lb);
        LCTY_HIP(hipMemcpyAsync(ws.init_plan.p, plan.groups.data(), gb, hipMemcpyHostToDevice, s));
        LCTY_HIP(hipMemcpyAsync(ws.init_plan.p + gb, plan.chains.data(), cb, hipMemcpyHostToDevice, s));
        if (plan.lds > 48 * 1024)
            LCTY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(solve_init_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         static_cast<int>(plan.lds)));
        const InitGroup* d_groups = reinterpret_cast<const InitGroup*>(ws.init_plan.p);
        const InitChainP* d_chains = reinterpret_cast<const InitChainP*>(ws.init_plan.p + gb);
        ctx->timed(V.solver.kind == LCTY_SOLVER_ANNEAL ? LCTY_K_SOLVE_INIT_ANNEAL : LCTY_K_SOLVE_INIT, [&] {
            hipLaunchKernelGGL(solve_init_tile_kernel, dim3(static_cast<uint32_t>(plan.groups.size())), dim3(256), plan.lds, s, V, d_groups, d_chains,
                               reinterpret_cast<unsigned long long*>(ws.init_plan.p + gb + cb), plan.T, plan.R);
        }, s);
        LCTY_HIP(hipGetLastError());
        return;
    }
    switch (V.ploidy) {
        case 1: launch_init_p<1>(ctx, V, nch, lds_init, s); break;
        case 2: launch_init_p<2>(ctx, V, nch, lds_init, s); break;
        case 3: launch_init_p<3>(ctx, V, nch, lds_init, s); break;
        default: launch_init_p<4>(ctx, V, nch, lds_init, s); break;
    }
}
void launch_pause(hipStream_t s) { hipLaunchKernelGGL(pause_kernel, dim3(1), dim3(64), 0, s, 8u); }
void launch_store_cur(ChainRec* recs, const uint32_t* words, uint64_t n, hipStream_t s) {
    hipLaunchKernelGGL(store_cur_kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, s, recs, words, n);
    LCTY_HIP(hipGetLastError());
}
void launch_pack_rows_count(uint64_t n, const LocCell* table, uint64_t ngp, uint32_t n_good, const uint16_t* alleles, uint32_t n_rows,
                            unsigned long long* total, hipStream_t s) {
    hipLaunchKernelGGL(pack_rows_count_kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, s, table, ngp, n_good, alleles, n_rows, total);
    LCTY_HIP(hipGetLastError());
}
void launch_pack_rows(uint64_t n, const LocCell* table, const uint32_t* table_ext, const double* table_unm, uint64_t ngp, uint32_t n_good,
                      const uint16_t* alleles, uint32_t n_rows, const PairAlnDev* pa, LocEntry* cells, uint64_t out_stride, PairAlnDev* extras,
                      unsigned long long* cursor, hipStream_t s) {
    hipLaunchKernelGGL(pack_rows_kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, s, table, table_ext, table_unm, ngp, n_good, alleles,
                       n_rows, pa, cells, out_stride, extras, cursor);
    LCTY_HIP(hipGetLastError());
}
void launch_place_rows(uint64_t n, const LocEntry* cells, uint64_t in_stride, uint32_t n_good, uint32_t n_rows, uint32_t ext_base, LocCell* full,
                       uint32_t* full_ext, double* full_unm, bool write_unm, uint64_t full_stride, uint64_t first, hipStream_t s) {
    hipLaunchKernelGGL(place_rows_kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, s, cells, in_stride, n_good, n_rows, ext_base, full,
                       full_ext, full_unm, write_unm, full_stride, first);
    LCTY_HIP(hipGetLastError());
}
void launch_pad_rows(uint64_t n, LocCell* full, uint32_t* full_ext, double* full_unm, uint64_t full_stride, uint64_t from, uint32_t n_rows, hipStream_t s) {
    hipLaunchKernelGGL(pad_rows_kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, s, full, full_ext, full_unm, full_stride, from, n_rows);
    LCTY_HIP(hipGetLastError());
}
void launch_count_unexplained(uint32_t blocks, const uint8_t* status, const double* unmapped, const double* matrix, uint64_t n_pairs, uint32_t A,
                              const uint16_t* ids, uint32_t ploidy, unsigned long long* out, hipStream_t s) {
    hipLaunchKernelGGL(count_unexplained_kernel, dim3(blocks), dim3(256), 0, s, status, unmapped, matrix, n_pairs, A, ids, ploidy, out);
    LCTY_HIP(hipGetLastError());
}
void launch_read_nw(const SolveView& V, uint32_t* nw, hipStream_t s) {
    if (!V.n_good) return;
    const uint32_t blocks = (V.n_good + 255) / 256;
    switch (V.ploidy) {
        case 1: hipLaunchKernelGGL(read_nw_kernel<1>, dim3(blocks), dim3(256), 0, s, V, nw); break;
        case 2: hipLaunchKernelGGL(read_nw_kernel<2>, dim3(blocks), dim3(256), 0, s, V, nw); break;
        case 3: hipLaunchKernelGGL(read_nw_kernel<3>, dim3(blocks), dim3(256), 0, s, V, nw); break;
        default: hipLaunchKernelGGL(read_nw_kernel<4>, dim3(blocks), dim3(256), 0, s, V, nw); break;
    }
    LCTY_HIP(hipGetLastError());
}
void launch_assignment_counts(const SolveView& V, const uint32_t* nw, const uint64_t* read_off, uint16_t* counts, hipStream_t s) {
    if (!V.n_good) return;
    hipLaunchKernelGGL(assignment_counts_kernel, dim3((V.n_good + 255) / 256), dim3(256), 0, s, V, nw, read_off, counts);
    LCTY_HIP(hipGetLastError());
}

}  // namespace lcty
