// lcty_solve_device.hpp — what the two halves of the solver stages share: the view of a stage that the kernels take by value
// (SolveView), the cells of the location table, the record list of a chain, the random streams (definitions shared with
// oracle/lcty_oracle_solve.c), and the launchers through which the host side starts the kernels.
//   lcty_solve_kernels.hip   the kernels and their launchers
//   lcty_solve_host.hip      the stage driver, the queue of loci, the final comparison, the C ABI
//   lcty_exact.cpp           the exact solver (host only)
#pragma once
#include "lcty_objects.hpp"

namespace lcty {

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

// ---- the initialisation of a diploid stage in GROUPS of chains (solve_init_tile_kernel) ----
// The 5 000 chains of a stage are pairs out of a few hundred alleles: a workgroup that builds the records of several chains whose
// genotypes share alleles reads every shared row of the location table once. The host cuts the stage into groups (plan_init_groups):
// at most INIT_TILE_T chains on at most INIT_TILE_R distinct rows; the attempts of a genotype share both rows.
constexpr uint32_t INIT_TILE_T = 8, INIT_TILE_R = 6;
constexpr uint32_t INIT_TILE_Q = 1024;      // places in a wavefront's list of deferred reads (a ring; a block adds at most 64 per chain to < 64)
static_assert(INIT_TILE_Q >= 64 * (INIT_TILE_T + 1) && (INIT_TILE_Q & (INIT_TILE_Q - 1)) == 0, "the list holds a block's entries");
struct __attribute__((aligned(16))) InitChainP {     // one chain of a group: GenotypeWindows (windows.rs:709-739) of its two contigs, made on the host
    uint64_t seed;
    uint32_t chain;                 // index of the chain in the batch (genotype * attempts + attempt)
    uint32_t ia, ib;                // which of the group's rows hold its two contigs
    uint32_t row0, row1;            // the same as rows of the location table
    uint32_t id0, id1;              // the alleles
    uint32_t shift0, rs0, re0, shift1, rs1, re1;     // first window, region start, region end of either contig
    uint32_t total_w;
};
static_assert(sizeof(InitChainP) == 64, "InitChainP layout");
struct __attribute__((aligned(16))) InitGroup { uint32_t first, n_chains, n_rows, _pad; uint32_t row[8]; };
static_assert(sizeof(InitGroup) == 48 && INIT_TILE_R <= 8, "InitGroup layout");
struct InitPlan {
    std::vector<InitGroup> groups; std::vector<InitChainP> chains;
    uint32_t T = 0, R = 0;          // the launch's largest group
    size_t lds = 0;
};

// ---- launchers of lcty_solve_kernels.hip (the only way the host side starts a solver kernel) ----
void ensure_solver_tables(lcty_reads* reads);                              // location table + compact "unmapped" column of a scored batch
void ensure_depth_table(lcty_locus* loc, uint64_t want);                   // extended depth table of the locus, at least `want` wide
void build_depth_table_into(const lcty_locus* loc, uint32_t depth, DevBuf<double>& out, hipStream_t s);   // the same table into a buffer of the caller's
bool solver_lds_fits(uint32_t wstride);                                    // the window arrays of one chain next to the annealing ring in 160 KB of LDS
// the batch of a stage as the host has it: what plan_init_groups makes the groups of a diploid stage from (row_of: NULL = the allele itself)
struct InitHost {
    const uint16_t* genotypes; const uint64_t* seeds; const uint16_t* row_of; const lcty_locus* loc; uint32_t n_cus;
    uint32_t lds_budget;       // bytes of LDS a group may take (0: half a CU's); the last stage of a locus in a queue starts beside the greedy
                               // chains of the next locus, whose two workgroups per CU leave 35 KB
};
void plan_init_groups(const SolveView& V, const InitHost& H, uint32_t nch, InitPlan& plan);
void launch_init(lcty_ctx* ctx, const SolveView& V, uint32_t nch, size_t lds_init, hipStream_t s, const InitHost* host, lcty_ctx::SolveWorkspace& ws,
                 InitPlan& plan);
void launch_greedy_chains(lcty_ctx* ctx, SolveView& V, uint32_t nch, hipStream_t stream, lcty_ctx::SolveWorkspace& ws);
void launch_anneal(lcty_ctx* ctx, const SolveView& V, uint32_t nch, hipStream_t s);
void launch_pause(hipStream_t s);
void launch_store_cur(ChainRec* recs, const uint32_t* words, uint64_t n, hipStream_t s);
void launch_pack_rows_count(uint64_t n, const LocCell* table, uint64_t ngp, uint32_t n_good, const uint16_t* alleles, uint32_t n_rows,
                            unsigned long long* total, hipStream_t s);
void launch_pack_rows(uint64_t n, const LocCell* table, const uint32_t* table_ext, const double* table_unm, uint64_t ngp, uint32_t n_good,
                      const uint16_t* alleles, uint32_t n_rows, const PairAlnDev* pa, LocEntry* cells, uint64_t out_stride, PairAlnDev* extras,
                      unsigned long long* cursor, hipStream_t s);
void launch_place_rows(uint64_t n, const LocEntry* cells, uint64_t in_stride, uint32_t n_good, uint32_t n_rows, uint32_t ext_base, LocCell* full,
                       uint32_t* full_ext, double* full_unm, bool write_unm, uint64_t full_stride, uint64_t first, hipStream_t s);
void launch_pad_rows(uint64_t n, LocCell* full, uint32_t* full_ext, double* full_unm, uint64_t full_stride, uint64_t from, uint32_t n_rows, hipStream_t s);
void launch_count_unexplained(uint32_t blocks, const uint8_t* status, const double* unmapped, const double* matrix, uint64_t n_pairs, uint32_t A,
                              const uint16_t* ids, uint32_t ploidy, unsigned long long* out, hipStream_t s);
void launch_read_nw(const SolveView& V, uint32_t* nw, hipStream_t s);
void launch_assignment_counts(const SolveView& V, const uint32_t* nw, const uint64_t* read_off, uint16_t* counts, hipStream_t s);

}  // namespace lcty
