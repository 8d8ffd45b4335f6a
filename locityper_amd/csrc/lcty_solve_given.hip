// lcty_solve_given.hip — `Solver::solve` on the GenotypeAlignments the CALLER holds (src/solvers/mod.rs:49-75
// as called at src/solvers/solve.rs:824-826: `gt_alns.apply_tweak(rng, ..); stage.solver.solve(&gt_alns, rng)`).
//
// lcty_solve_stage builds the chains of a stage itself from the scored reads of a batch and tweaks them from a seed. This entry point
// takes what the reference hands to a solver instead — the locations of every read pair on the genotype with the windows apply_tweak
// left them (assgn.rs:16-36, 127-151), the window distributions, the two contributions — and runs ONE chain of the chosen solver on
// it, on the device, with the loop kernels of the batched stages (lcty_solve_kernels.hip): given_init_kernel turns the CSR into the
// chain's record list, window arrays and depth histogram (what solve_init_kernel makes from the location table), the greedy /
// annealing loop runs with one chain, given_assign_kernel reads the assignment back. The exact solver takes the model on the host
// (lcty_exact.cpp), as in the batched stage.
//
// Re-entrant: `Solver::solve` is called from the reference's worker threads at once, one call per (genotype, attempt)
// (solve.rs:1010-1017, 1124-1125). A call takes a slot of the context — a stream, a chain workspace, a depth table of its own — or
// makes one; nothing of the locus is written.
#include <algorithm>
#include <cmath>
#include <memory>

#include "lcty_exact.hpp"
#include "lcty_solve_device.hpp"

using namespace lcty;

namespace lcty {

struct GivenDev {
    const uint64_t* read_ixs;       // [n_reads + 1]
    const double* lp;               // [n_alns]
    const uint32_t* win;            // [2 * n_alns]
    const uint8_t* gc;              // [n_windows]
    const double* weight;           // [n_windows]
    uint32_t n_reads, n_windows;
    uint64_t seed;
    uint32_t random_start;
};

// K12 + K13 of one chain from the caller's arrays: window distributions as given; initial assignment (ReadAssignment::try_new,
// assgn.rs:199-226: location 0, or a draw per non-trivial read), depth histogram, the records of the non-trivial reads in read order
// (assgn.rs:61-63). One workgroup; wavefront k compacts the k-th range of the reads into the k-th part of the record list (RecList).
__global__ __launch_bounds__(256) void given_init_kernel(const SolveView V, const GivenDev G) {
    extern __shared__ __align__(16) uint8_t smem[];
    uint32_t* depth = reinterpret_cast<uint32_t*>(smem);                          // [wstride]
    double* red = reinterpret_cast<double*>(smem + ((static_cast<size_t>(V.wstride) * 4 + 15) & ~static_cast<size_t>(15)));   // [256]
    uint32_t* seg_cnt = reinterpret_cast<uint32_t*>(red + 256);                   // [4] records of every segment, [4] = further locations handed out
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t w = tid; w < G.n_windows; w += 256) {
        depth[w] = 0;
        const double weight = G.weight[w];
        V.c_ww[w] = weight;                                                       // 0 = WindowDistr::TRIVIAL (distr_cache.rs:28-31)
        V.c_gc[w] = weight != 0.0 ? G.gc[w] : static_cast<uint8_t>(0);
    }
    if (tid == 0) seg_cnt[4] = 0;
    __syncthreads();
    ChainRec* recs = V.recs + static_cast<uint64_t>(wave) * V.seg_reads;
    const uint32_t seg_lo = min(wave * V.seg_reads, G.n_reads), seg_hi = min(seg_lo + V.seg_reads, G.n_reads);
    double aln_part = 0.0;
    uint32_t n_recs = 0;
    for (uint32_t base = seg_lo; base < seg_hi; base += 64) {
        const uint32_t rp = base + lane;
        uint32_t nloc = 0, a0 = 0;
        uint64_t i0 = 0;
        if (rp < seg_hi) {
            i0 = G.read_ixs[rp];
            nloc = static_cast<uint32_t>(G.read_ixs[rp + 1] - i0);
            if (nloc > 1 && G.random_start)                                       // rng.random_range(0..alns.len()), our adaptor (oracle/lcty_oracle.h)
                a0 = static_cast<uint32_t>(__umul64hi(counter_u64(G.seed ^ INIT_KEY_XOR, rp), static_cast<uint64_t>(nloc)));
        }
        const bool nontrivial = nloc > 1;
        const unsigned long long nt_mask = __ballot(nontrivial);
        const uint32_t slot = n_recs + static_cast<uint32_t>(__popcll(nt_mask & ((1ull << lane) - 1ull)));
        n_recs += static_cast<uint32_t>(__popcll(nt_mask));
        if (nloc > 0) {
            const uint32_t n_extra = nloc > 2 ? nloc - 2u : 0u;
            uint32_t eix = 0;
            if (n_extra) eix = atomicAdd(&seg_cnt[4], n_extra);
            ChainRec rec; rec.rp_cur = rp | (a0 << 24); rec.meta = nloc | (eix << 8);
            rec.lp0 = rec.lp1 = 0.0; rec.win0 = rec.win1 = 0;
            for (uint32_t t = 0; t < nloc; t++) {
                const double lp = G.lp[i0 + t];
                const uint32_t wa = G.win[2 * (i0 + t)], wb = G.win[2 * (i0 + t) + 1];
                const uint32_t win = wa | (wb << 16);
                if (t == 0) { rec.lp0 = lp; rec.win0 = win; }
                else if (t == 1) { rec.lp1 = lp; rec.win1 = win; }
                else if (eix + t - 2 < V.extra_cap) { ExtraLoc e; e.lp = lp; e.win = win; e._pad = 0; V.extra[eix + t - 2] = e; }
                if (t == a0) {
                    atomicAdd(&depth[wa], 1u);
                    atomicAdd(&depth[wb], 1u);
                    aln_part += lp;
                }
            }
            if (nontrivial) recs[slot] = rec;
        }
    }
    if (lane == 0) seg_cnt[wave] = n_recs;
    red[tid] = aln_part;
    __syncthreads();
    if (tid == 0 && seg_cnt[4] > V.extra_cap) atomicMax(V.overflow, 4u);
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    for (uint32_t w = tid; w < G.n_windows; w += 256) V.c_depth[w] = depth[w];
    if (tid == 0) {
        uint32_t run = 0;
        for (uint32_t k = 0; k < INIT_SEGS; k++) { V.c_seg[k] = run; run += seg_cnt[k]; }
        V.c_aln[0] = red[0]; V.c_nnt[0] = run; V.c_totw[0] = G.n_windows;
    }
}

// read_assgn of the chain (assgn.rs:176-177): the location byte of every record; trivial reads stay at 0
__global__ __launch_bounds__(256) void given_assign_kernel(const SolveView V, uint16_t* __restrict__ assgn) {
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= V.c_nnt[0]) return;
    const RecList recs{V.recs, V.c_seg, V.seg_reads};
    const uint32_t w = recs[s].rp_cur;
    assgn[w & 0xFFFFFFu] = static_cast<uint16_t>(w >> 24);
}

}  // namespace lcty

namespace {

struct SlotLease {                       // a slot of the context for the lifetime of one call
    lcty_ctx* ctx; lcty_ctx::GivenSlot* slot;
    explicit SlotLease(lcty_ctx* c) : ctx(c), slot(nullptr) {
        std::lock_guard<std::mutex> g(ctx->given_mutex);
        for (auto& s : ctx->given_slots) if (!s->busy) { slot = s.get(); break; }
        if (!slot) {
            std::unique_ptr<lcty_ctx::GivenSlot> fresh(new lcty_ctx::GivenSlot());
            LCTY_HIP(hipStreamCreateWithFlags(&fresh->stream, hipStreamNonBlocking));
            ctx->given_slots.push_back(std::move(fresh));
            slot = ctx->given_slots.back().get();
        }
        slot->busy = true;
    }
    ~SlotLease() { std::lock_guard<std::mutex> g(ctx->given_mutex); slot->busy = false; }
    SlotLease(const SlotLease&) = delete; SlotLease& operator=(const SlotLease&) = delete;
};

inline uint64_t rotl64(uint64_t v, int k) { return (v << k) | (v >> (64 - k)); }
uint64_t xoshiro_next(uint64_t* s) {                            // xoshiro256++ (rand_xoshiro::Xoshiro256PlusPlus::next_u64)
    const uint64_t result = rotl64(s[0] + s[3], 23) + s[0];
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl64(s[3], 45);
    return result;
}

void check_solver(const lcty_solver* solver) {
    if (solver->kind != LCTY_SOLVER_GREEDY && solver->kind != LCTY_SOLVER_ANNEAL && solver->kind != LCTY_SOLVER_EXACT)
        fail(LCTY_ERR_INVALID_INPUT, "unknown solver kind");
    if (solver->kind == LCTY_SOLVER_ANNEAL && !(solver->init_prob > 0.0 && solver->init_prob <= 1.0))
        fail(LCTY_ERR_INVALID_INPUT, "Initial probability (%g) must be within (0, 1]", solver->init_prob);
    if (solver->kind == LCTY_SOLVER_ANNEAL && solver->anneal_steps == 0) fail(LCTY_ERR_INVALID_INPUT, "Number of annealing steps must be positive");
    if (solver->kind == LCTY_SOLVER_GREEDY && solver->sample_size == 0) fail(LCTY_ERR_INVALID_INPUT, "Sample size must be positive");
    if (solver->kind == LCTY_SOLVER_GREEDY && solver->sample_size > 64) fail(LCTY_ERR_UNSUPPORTED, "greedy sample size above 64");
}

// what every location that names a window adds up to: no assignment can make the window deeper (a pair in one window counts twice)
uint64_t deepest_window(const lcty_gt_alns_view* g, std::vector<uint32_t>& reach) {
    const uint32_t W = g->n_windows;
    const uint64_t n_alns = g->read_ixs[g->n_reads];
    reach.assign(W, 0);
    for (uint64_t i = 0; i < n_alns; i++) {
        const uint32_t wa = g->windows[2 * i], wb = g->windows[2 * i + 1];
        if (wa >= W || wb >= W) fail(LCTY_ERR_INVALID_INPUT, "location %llu lies in window %u of %u", static_cast<unsigned long long>(i), std::max(wa, wb), W);
        reach[wa]++; reach[wb]++;
    }
    uint64_t deepest = 0;
    for (uint32_t w = 0; w < W; w++) if (g->window_weight[w] != 0.0) deepest = std::max<uint64_t>(deepest, reach[w]);
    return deepest;
}

void check_view(const lcty_gt_alns_view* g) {
    if (!g) fail(LCTY_ERR_INVALID_INPUT, "null argument");
    if (!g->read_ixs || (g->n_windows && (!g->window_gc || !g->window_weight))) fail(LCTY_ERR_INVALID_INPUT, "null argument");
    if (g->n_windows < 2) fail(LCTY_ERR_INVALID_INPUT, "a genotype has at least the two windows of unmapped and out-of-region reads (windows.rs:70-76)");
    if (g->n_reads >= (1ull << 24)) fail(LCTY_ERR_UNSUPPORTED, "the device solver handles up to 2^24 read pairs per locus");
    if (g->read_ixs[0] != 0) fail(LCTY_ERR_INVALID_INPUT, "read_ixs[0] must be 0 (assgn.rs:29-31)");
    if (g->read_ixs[g->n_reads] && (!g->ln_prob || !g->windows)) fail(LCTY_ERR_INVALID_INPUT, "null argument");
    for (uint64_t r = 0; r < g->n_reads; r++)
        if (g->read_ixs[r + 1] <= g->read_ixs[r])
            fail(LCTY_ERR_INVALID_INPUT, "Read pair %llu has zero possible alignment locations", static_cast<unsigned long long>(r));
}

// `loc`: the window distributions are the locus' DistrCache rows, window_gc = GC bin. `T`: they are rows of the caller's table.
void solve_given(lcty_ctx* ctx, lcty_locus* loc, const lcty_depth_tables* T, const lcty_gt_alns_view* g, const lcty_solver* solver,
                 uint64_t* rng_state, uint16_t* read_assgn, double* lik_parts, double* likelihood) {
    if (!ctx || (!loc && !T) || !g || !solver || !read_assgn) fail(LCTY_ERR_INVALID_INPUT, "null argument");
    if (T && (!T->values || T->n_rows == 0 || T->n_rows > 128 || T->width < 2)) fail(LCTY_ERR_INVALID_INPUT, "depth tables: 1..128 rows of at least two depths");
    check_view(g);
    check_solver(solver);
    const uint64_t R = g->n_reads;
    const uint32_t W = g->n_windows;
    const uint64_t n_alns = g->read_ixs[R];
    if (g->n_contigs > 16 || (g->n_contigs && !g->wshifts)) fail(LCTY_ERR_INVALID_INPUT, "wshifts: n_contigs + 1 entries for at most 16 contigs");
    // what the reference asserts while it builds the object (assgn.rs:55-58), and what the records hold
    uint64_t nnt = 0, n_extra = 0;
    for (uint64_t r = 0; r < R; r++) {
        const uint64_t m = g->read_ixs[r + 1] - g->read_ixs[r];
        if (m > 255) fail(LCTY_ERR_UNSUPPORTED, "Read pair %llu has too many alignment locations (%llu): the device solver keeps a location in 8 bits",
                          static_cast<unsigned long long>(r), static_cast<unsigned long long>(m));
        if (m > 1) { nnt++; n_extra += m - 2; }
    }
    if (n_extra >= (1ull << 24)) fail(LCTY_ERR_UNSUPPORTED, "2^24 or more further locations in one chain");
    for (uint64_t i = 0; i < n_alns; i++)
        if (std::isnan(g->ln_prob[i])) fail(LCTY_ERR_INVALID_INPUT, "location %llu has no ln-probability", static_cast<unsigned long long>(i));
    const uint32_t n_rows = T ? T->n_rows : LCTY_GC_BINS;
    for (uint32_t w = 0; w < W; w++) {
        if (!(g->window_weight[w] >= 0.0)) fail(LCTY_ERR_INVALID_INPUT, "window %u has weight %g", w, g->window_weight[w]);
        if (g->window_weight[w] != 0.0 && g->window_gc[w] >= n_rows)
            fail(LCTY_ERR_INVALID_INPUT, "window %u names distribution %u of %u", w, g->window_gc[w], n_rows);
    }
    std::vector<uint32_t> reach;
    const uint64_t deepest = deepest_window(g, reach);
    if (T && deepest + 1 > T->width)
        fail(LCTY_ERR_INVALID_INPUT, "depth tables of %u depths: a window of this genotype can get %llu deep (lcty_gt_alns_deepest)", T->width,
             static_cast<unsigned long long>(deepest));

    // Solver::solve (mod.rs:59-72): without non-trivial reads there is one assignment and the generator is not touched
    uint64_t seed = 0;
    if (nnt) {
        if (!rng_state) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        seed = xoshiro_next(rng_state);                          // the chain's seed: one draw of the caller's generator, as a stage takes one per chain
    }

    ctx->activate();
    SlotLease lease(ctx);
    lcty_ctx::GivenSlot& S = *lease.slot;
    hipStream_t s = S.stream;
    auto& ws = S.ws;

    SolveView V{};
    V.depth_contrib = g->depth_contrib; V.aln_contrib = g->aln_contrib;
    V.n_wk = V.n_wc = 0;                                          // weights as given: no tables
    V.n_good = static_cast<uint32_t>(R);
    V.seg_reads = static_cast<uint32_t>(((R + INIT_SEGS - 1) / INIT_SEGS + 63) / 64 * 64);
    if (V.seg_reads == 0) V.seg_reads = 64;
    V.rstride = static_cast<uint64_t>(INIT_SEGS) * V.seg_reads;
    V.ploidy = g->n_contigs ? g->n_contigs : 1; V.attempts = 1; V.solver = *solver; V.priors = nullptr;
    V.wstride = (W + 3) & ~3u;
    if (!solver_lds_fits(V.wstride)) fail(LCTY_ERR_UNSUPPORTED, "%u windows per genotype: too many for the device solver", W);
    // the slot's depth table, wide enough for the deepest window: DistrCache (distr_cache.rs:61-75) of the locus, made on the device —
    // or the caller's rows (a power-of-two row stride; what lies behind a row's `width` is never read: deepest < width)
    if (loc) {
        uint32_t depth = LCTY_DEPTH_CACHE;
        while (depth < deepest + 2) depth *= 2;
        if (S.lut_of != loc->serial || S.lut_is_given || S.lut_depth < depth) {
            build_depth_table_into(loc, depth, S.lut, s);
            S.lut_of = loc->serial; S.lut_is_given = false; S.lut_depth = depth;
        }
    } else if (!(S.lut_is_given && T->id != 0 && S.lut_of == T->id && S.lut_given_width == T->width && S.lut_given_rows == T->n_rows)) {
        uint32_t depth = 2;
        while (depth < T->width) depth *= 2;
        if (depth > (1u << 22)) fail(LCTY_ERR_UNSUPPORTED, "depth tables more than 4 M depths wide");
        S.lut.ensure(static_cast<size_t>(128) * depth);
        LCTY_HIP(hipMemcpy2DAsync(S.lut.p, static_cast<size_t>(depth) * 8, T->values, static_cast<size_t>(T->width) * 8, static_cast<size_t>(T->width) * 8,
                                  T->n_rows, hipMemcpyHostToDevice, s));
        S.lut_of = T->id; S.lut_is_given = true; S.lut_depth = depth; S.lut_given_width = T->width; S.lut_given_rows = T->n_rows;
    }
    V.lut = S.lut.p; V.lut_depth = S.lut_depth; V.lut_shift = static_cast<uint32_t>(__builtin_ctz(S.lut_depth));
    const uint32_t extra_cap = static_cast<uint32_t>(n_extra) + 2;     // two spare entries: the greedy loop reads a pair per record
    ws.recs.ensure_slack(V.rstride); ws.extra.ensure_slack(extra_cap);
    ws.cww.ensure_slack(V.wstride); ws.cgc.ensure_slack(V.wstride); ws.cdepth.ensure_slack(V.wstride);
    ws.cnnt.ensure(1); ws.cseg.ensure(4); ws.ctotw.ensure(1); ws.caln.ensure(1); ws.seeds.ensure(1); ws.liks.ensure(1); ws.parts.ensure(4);
    ws.ovf.ensure(2); ws.ovf.zero(s);
    V.seeds = ws.seeds.p; V.recs = ws.recs.p; V.extra = ws.extra.p; V.extra_cap = extra_cap; V.liks = ws.liks.p; V.parts = ws.parts.p;
    V.c_ww = ws.cww.p; V.c_uc = nullptr; V.c_gc = ws.cgc.p; V.c_depth = ws.cdepth.p; V.c_nnt = ws.cnnt.p; V.c_seg = ws.cseg.p;
    V.c_totw = ws.ctotw.p; V.c_aln = ws.caln.p; V.overflow = ws.ovf.p; V.dbg = nullptr;
    ws.seeds.upload(&seed, 1, s);

    S.read_ixs.ensure_slack(R + 1); S.lp.ensure_slack(n_alns); S.win.ensure_slack(2 * n_alns); S.gc.ensure_slack(W); S.weight.ensure_slack(W);
    S.assgn.ensure_slack(R);
    S.read_ixs.upload(g->read_ixs, R + 1, s); S.lp.upload(g->ln_prob, n_alns, s); S.win.upload(g->windows, 2 * n_alns, s);
    S.gc.upload(g->window_gc, W, s); S.weight.upload(g->window_weight, W, s);
    const bool random_start = solver->kind == LCTY_SOLVER_ANNEAL || (solver->kind == LCTY_SOLVER_GREEDY && !solver->best_start);
    const GivenDev G{S.read_ixs.p, S.lp.p, S.win.p, S.gc.p, S.weight.p, static_cast<uint32_t>(R), W, seed, random_start ? 1u : 0u};
    const size_t lds_init = ((static_cast<size_t>(V.wstride) * 4 + 15) & ~static_cast<size_t>(15)) + 256 * 8 + 64;
    if (lds_init > 48 * 1024)
        LCTY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(given_init_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_init)));
    hipLaunchKernelGGL(given_init_kernel, dim3(1), dim3(256), lds_init, s, V, G);
    LCTY_HIP(hipGetLastError());

    double parts[4] = {0, 0, 0, 0}, lik = 0.0;
    if (solver->kind == LCTY_SOLVER_EXACT && nnt) {
        // the integer programme of highs.rs:38-100 on the host (lcty_exact.cpp), from the caller's arrays
        exact::Model m;
        m.n = static_cast<uint32_t>(nnt); m.tw = W;
        m.ww.assign(g->window_weight, g->window_weight + W); m.gcb.resize(W);
        for (uint32_t w = 0; w < W; w++) m.gcb[w] = g->window_weight[w] != 0.0 ? g->window_gc[w] : 0;
        m.depth0.assign(W, 0); m.first.reserve(nnt + 1);
        std::vector<uint32_t> nt_read; nt_read.reserve(nnt);
        double aln0 = -0.0;
        for (uint64_t r = 0; r < R; r++) {
            const uint64_t i0 = g->read_ixs[r], n = g->read_ixs[r + 1] - i0;
            aln0 += g->ln_prob[i0];
            m.depth0[g->windows[2 * i0]]++; m.depth0[g->windows[2 * i0 + 1]]++;
            if (n > 1) {
                nt_read.push_back(static_cast<uint32_t>(r));
                m.first.push_back(static_cast<uint32_t>(m.locs.size()));
                for (uint64_t t = 0; t < n; t++) m.locs.push_back({g->ln_prob[i0 + t], g->windows[2 * (i0 + t)], g->windows[2 * (i0 + t) + 1]});
            }
        }
        m.first.push_back(static_cast<uint32_t>(m.locs.size()));
        m.aln0 = aln0;
        if (g->n_contigs) m.allele_first_w.assign(g->wshifts, g->wshifts + g->n_contigs + 1);
        else { m.allele_first_w.assign(1, 2u); m.allele_first_w.push_back(W); }
        m.aln_contrib = g->aln_contrib; m.depth_contrib = g->depth_contrib;
        m.node_limit = solver->node_limit ? solver->node_limit : 20ull * 1000 * 1000;
        m.rel_gap = solver->init_prob > 0.0 && solver->init_prob < 1.0 ? solver->init_prob : 0.0;
        m.gc_bins = n_rows;
        std::vector<double> lut(static_cast<size_t>(n_rows) * S.lut_depth);
        S.lut.download(lut.data(), lut.size(), s);
        LCTY_HIP(hipStreamSynchronize(s));
        // (no window with a distribution gets deeper than `deepest` < the table's width: the search indexes the table for those only)
        exact::Result res;
        exact::solve(m, lut.data(), S.lut_depth, res);
        if (res.out_of_nodes)
            fail(LCTY_ERR_SOLVER, "Exact solver: no proof of optimality within %llu nodes (%u non-trivial reads, %u of them free after fixing the dominated ones); Model finished with non-optimal status NodeLimit",
                 static_cast<unsigned long long>(m.node_limit), m.n, res.n_free);
        memset(read_assgn, 0, sizeof(uint16_t) * R);
        for (uint32_t i = 0; i < m.n; i++) read_assgn[nt_read[i]] = res.assign[i];
        parts[0] = res.aln_lik; parts[1] = res.depth_lik; lik = res.value;
    } else {
        // trivial genotypes take the greedy kernel's prologue: likelihood of the one assignment (recalc_likelihood, assgn.rs:346-354)
        if (solver->kind == LCTY_SOLVER_ANNEAL && nnt) launch_anneal(ctx, V, 1, s);
        else launch_greedy_chains(ctx, V, 1, s, ws);
        LCTY_HIP(hipMemsetAsync(S.assgn.p, 0, sizeof(uint16_t) * std::max<uint64_t>(R, 1), s));
        if (nnt) {
            hipLaunchKernelGGL(given_assign_kernel, dim3(static_cast<uint32_t>((nnt + 255) / 256)), dim3(256), 0, s, V, S.assgn.p);
            LCTY_HIP(hipGetLastError());
        }
        uint32_t ovf[2] = {0, 0};
        ws.ovf.download(ovf, 2, s);
        ws.parts.download(parts, 4, s); ws.liks.download(&lik, 1, s);
        S.assgn.download(read_assgn, R, s);
        LCTY_HIP(hipStreamSynchronize(s));
        if (ovf[0] == 3) fail(LCTY_ERR_RUNTIME, "annealing kernel: the staging wavefront and the chain lost each other");
        if (ovf[0]) fail(LCTY_ERR_RUNTIME, "solver chain: flag %u (a window deeper than the reads that can reach it, or more locations than counted)", ovf[0]);
    }
    if (lik_parts) { lik_parts[0] = parts[0]; lik_parts[1] = parts[1]; }
    if (likelihood) *likelihood = lik;
}

}  // namespace

extern "C" {

int32_t lcty_solve_given(lcty_locus* locus, const lcty_gt_alns_view* gt_alns, const lcty_solver* solver, uint64_t* rng_state,
                         uint16_t* read_assgn, double* lik_parts, double* likelihood) {
    return guarded([&] {
        if (!locus) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        solve_given(locus->ctx, locus, nullptr, gt_alns, solver, rng_state, read_assgn, lik_parts, likelihood);
    });
}

int32_t lcty_solve_given_tables(lcty_ctx* ctx, const lcty_gt_alns_view* gt_alns, const lcty_depth_tables* tables, const lcty_solver* solver,
                                uint64_t* rng_state, uint16_t* read_assgn, double* lik_parts, double* likelihood) {
    return guarded([&] {
        if (!tables) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        solve_given(ctx, nullptr, tables, gt_alns, solver, rng_state, read_assgn, lik_parts, likelihood);
    });
}

int32_t lcty_gt_alns_deepest(const lcty_gt_alns_view* gt_alns, uint32_t* deepest) {
    return guarded([&] {
        if (!deepest) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        check_view(gt_alns);
        std::vector<uint32_t> reach;
        *deepest = static_cast<uint32_t>(deepest_window(gt_alns, reach));
    });
}

// XoshiroRng::seed_from_u64 (ext/rand.rs:3-22: SplitMix64 fill) / next_u64 for callers that keep the generator's four words themselves
int32_t lcty_rng_seed_from_u64(uint64_t seed, uint64_t* state) {
    return guarded([&] {
        if (!state) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        uint64_t x = seed;
        for (int i = 0; i < 4; i++) {
            uint64_t z = (x += 0x9e3779b97f4a7c15ull);
            z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
            z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
            state[i] = z ^ (z >> 31);
        }
    });
}
int32_t lcty_rng_next_u64(uint64_t* state, uint64_t* out) {
    return guarded([&] {
        if (!state || !out) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        *out = xoshiro_next(state);
    });
}

}  // extern "C"
