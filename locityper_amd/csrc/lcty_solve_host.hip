// lcty_solve_host.hip — the host side of the solver stages: one stage over its (genotype, attempt) chains in
// batches that fit the device (StageRunner), the exact solver's models to and from the host (lcty_exact.cpp), the stages of a locus, the queue
// of loci with the last stage of a locus beside the next locus, rows of the location table between read shards (RowGatherer), the final
// comparison (K15) and the C ABI of all of it. The kernels and their launchers are in lcty_solve_kernels.hip.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <exception>
#include <memory>
#include <numeric>
#include <future>
#include <thread>
#include <type_traits>

#include "lcty_exact.hpp"
#include "lcty_solve_device.hpp"

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
    InitPlan init_plan;                 // the groups of the batch being initialised (host copy: its upload is asynchronous)
    InitHost init_host{nullptr, nullptr, nullptr, nullptr, 0, 0};
    const RowGatherer* gathered_rows = nullptr;

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
        ctx = reads->ctx; loc = reads->locus; gathered_rows = gathered;
        ctx->activate();
        stream = lane ? ctx->side_stream() : ctx->stream;
        reads->check_device_error(stream);
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
        lds_init = ((static_cast<size_t>(V.wstride) * 4 + 15) & ~static_cast<size_t>(15)) + 256 * 8 + 64;
        if (!solver_lds_fits(V.wstride))
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
            if (ctx->diag_knob("queue_trace", 0))
                fprintf(stderr, "[lcty queue] lane %u workspace: %llu chains x %llu places (had %.1f GB of records, %.1f GB of runs; free %.1f GB, budget %.1f GB, %u further locations per chain)\n",
                        lane, static_cast<unsigned long long>(max_chains), static_cast<unsigned long long>(ngp), ws.recs.n * 32e-9, ws.extra.n * 16e-9,
                        free_b * 1e-9, budget * 1e-9, ws.extra_cap);
            // Grow-only, with a little head-room: the loci of a queue differ by a fraction of a per cent in their good read pairs, and a
            // workspace that followed every locus exactly was released and allocated again (4 s for 150 GB, with every stream of the
            // device waiting) whenever a slightly larger locus came after a smaller one.
            uint64_t want_recs = std::max<uint64_t>(ws.recs.n, max_chains * (ngp + ngp / 128));
            uint64_t want_extra = std::max<uint64_t>(ws.extra.n, max_chains * static_cast<uint64_t>(ws.extra_cap) + 2);   // two spare entries: the greedy loop reads a pair per record
            if (want_recs * sizeof(ChainRec) + want_extra * sizeof(ExtraLoc) > budget) {
                // ... unless what is kept does not fit beside what is needed (longer runs of further locations for fewer chains): exactly then
                want_recs = max_chains * (ngp + ngp / 128); want_extra = max_chains * static_cast<uint64_t>(ws.extra_cap) + 2;
            }
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
        if (lane == 0) wait_for_tail_init();
        launch_init(ctx, V, nch, lds_init, stream, init_host.genotypes ? &init_host : nullptr, ws, init_plan);
        if (lane == 1) announce_tail_init();
        if (V.solver.kind == LCTY_SOLVER_EXACT) { solve_exact_batch(nch); return; }
        if (V.solver.kind == LCTY_SOLVER_ANNEAL) {
            // (the stage's initialisation above does NOT wait: beside the greedy loop — the CU's L1 path full of its gathers — it took 216 ms
            // instead of 8; it runs beside the next locus' initialisation, in groups small enough to fit next to that one's: InitHost::lds_budget)
            if (lane == 1) wait_for_greedy_of_next_locus();
            const bool timed_anneal = ctx->diag_knob("solve_anneal_timing", 0) != 0;
            V.dbg = nullptr;
            if (timed_anneal) { ws.dbg.ensure(12 * static_cast<size_t>(nch)); ws.dbg.zero(stream); V.dbg = ws.dbg.p; }
            launch_anneal(ctx, V, nch, stream);
            if (timed_anneal) {
                // diagnostic: the second loop of the annealing chains (stoch.rs:228-241), shader-clock ticks per ROUND and phase, means over the chains
                std::vector<double> d(12 * static_cast<size_t>(nch));
                ws.dbg.download(d.data(), d.size(), stream);
                LCTY_HIP(hipStreamSynchronize(stream));
                double sum[10] = {0}; size_t used = 0;
                for (size_t c = 0; c < nch; c++) if (d[12 * c + 6] > 0) { used++; for (int k = 0; k < 6; k++) sum[k] += d[12 * c + k] / d[12 * c + 6]; for (int k = 6; k < 10; k++) sum[k] += d[12 * c + k]; }
                if (used) fprintf(stderr, "[lcty anneal phases] %zu chains; second loop: %.0f rounds, %.1f moves per round, accepted %.0f of %.0f moves in all; ticks per round: wait for the ring + words %.0f, "
                                          "moves from the ring %.0f, depths + gathers + score %.0f, ballots + chain walk %.0f, apply + retire %.0f; loop total per round %.0f\n",
                                  used, sum[6] / used, sum[7] / std::max(sum[6], 1.0), sum[8] / used, sum[9] / used, sum[0] / used, sum[1] / used, sum[2] / used, sum[3] / used, sum[4] / used, sum[5] / used);
            }
            if (ctx->diag_knob("queue_trace", 0)) fprintf(stderr, "[lcty queue] %.3f ms batch %p annealing launched (lane %u)\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(), static_cast<const void*>(reads), lane);
            return;
        }
        if (lane == 0) announce_greedy();
        if (ctx->diag_knob("queue_trace", 0)) fprintf(stderr, "[lcty queue] %.3f ms batch %p greedy about to launch (lane %u)\n",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(), static_cast<const void*>(reads), lane);
        launch_greedy_chains(ctx, V, nch, stream, ws);
    }

    // lcty_ctx::LaunchGate: the side stream's initialisation (the last stage of the locus before) goes first, the main stream's behind it
    void announce_tail_init() {
        auto& g = ctx->gate;
        {
            std::lock_guard<std::mutex> lock(g.m);
            if (!g.init_ev) LCTY_HIP(hipEventCreateWithFlags(&g.init_ev, hipEventDisableTiming));
            LCTY_HIP(hipEventRecord(g.init_ev, stream));
            g.tail_inits = g.tails_started;
        }
        g.cv.notify_all();
    }
    void wait_for_tail_init() {
        auto& g = ctx->gate;
        std::unique_lock<std::mutex> lock(g.m);
        g.cv.wait(lock, [&] { return g.tail_inits >= g.tails_started; });      // a tail that has just been started: until it has issued its initialisation
        if (g.init_ev) LCTY_HIP(hipStreamWaitEvent(stream, g.init_ev, 0));      // (an older one's event has long been reached)
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
            launch_pause(stream);
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
        // the pool: `exact_threads`, by default what `host_threads` gives this context (bench.py --gpus N: cores / N per rank), at most 64; the head
        // lane and the tail thread of a queue each have their pool, so two exact stages at once share the host's cores
        const int64_t host_share = ctx->knob("host_threads", 0) > 0 ? ctx->knob("host_threads", 0) : static_cast<int64_t>(hw);
        const uint32_t n_threads = static_cast<uint32_t>(std::max<int64_t>(1, std::min<int64_t>(ctx->knob("exact_threads", std::min<int64_t>(host_share, 64)), 256)));
        const int trace = static_cast<int>(ctx->diag_knob("exact_trace", 0));
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
            // an exception on a worker (std::bad_alloc on a large model) must not reach std::terminate: the first one is kept, the workers
            // stop taking models, and it is thrown again on the calling thread once they have joined
            std::exception_ptr failed;
            std::mutex failed_mutex;
            auto work = [&] {
                try {
                    for (size_t k; (k = next.fetch_add(1)) < gn;) exact::solve(held[k].model, lut.data(), ld, held[k].res);
                } catch (...) {
                    std::lock_guard<std::mutex> g(failed_mutex);
                    if (!failed) failed = std::current_exception();
                    next.store(gn);
                }
            };
            const uint32_t nt = static_cast<uint32_t>(std::min<size_t>(n_threads, gn));
            if (nt <= 1) work();
            else {
                std::vector<std::thread> pool;
                for (uint32_t t = 0; t < nt; t++) pool.emplace_back(work);
                for (auto& th : pool) th.join();
            }
            if (failed) std::rethrow_exception(failed);
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
                    launch_store_cur(V.recs + static_cast<uint64_t>(c) * V.rstride, d_words.p, V.rstride, s);
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
        for (uint64_t g0 = 0; g0 < n_gt;) {
            const uint64_t ng = std::min(gt_per_batch, n_gt - g0), nch = ng * attempts;
            bool smaller_batches = false;
            upload_genotypes(genotypes + g0 * ploidy, ng);
            ws.seeds.upload(chain_seeds + g0 * attempts, nch, s);
            init_host = InitHost{genotypes + g0 * ploidy, chain_seeds + g0 * attempts, gathered_rows ? gathered_rows->row_of.data() : nullptr, loc,
                                 static_cast<uint32_t>(ctx->props.multiProcessorCount), lane == 1 ? 32u * 1024u : 0u};
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
                    if (ctx->diag_knob("solve_stats", 0))
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
                    if (ctx->diag_knob("queue_trace", 0))
                        fprintf(stderr, "[lcty queue] lane %u: %u further locations per chain, room for %llu genotypes a batch (had %llu)\n", lane, ws.extra_cap,
                                static_cast<unsigned long long>(gt_per_batch), static_cast<unsigned long long>(before));
                    // the longer runs leave room for fewer chains than this batch holds: the batch again from its first genotype, in
                    // smaller batches (it used to end the call: a 1 M-read ONT locus whose batch shared the device with 22 GB of records)
                    if (gt_per_batch < ng) { smaller_batches = true; break; }
                    continue;
                }
                if (loc->lut_ext_depth >= depth_cap) fail(LCTY_ERR_RUNTIME, "window depth beyond 2 * reads + 2");
                ensure_depth_table(loc, std::min<uint64_t>(4ull * loc->lut_ext_depth, depth_cap));
            }
            if (smaller_batches) continue;
            after_batch(g0, ng, liks.data());
            g0 += ng;
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
        launch_pack_rows_count(n, reinterpret_cast<const LocCell*>(shard->d_loc_table.p), shard->ngp, static_cast<uint32_t>(shard->n_good_cached),
                               B.alleles.p, n_rows, B.counters.p + 2 * slot, stream);
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
    launch_pack_rows(n, reinterpret_cast<const LocCell*>(shard->d_loc_table.p), shard->d_loc_ext.p, shard->d_loc_unm.p, shard->ngp,
                     static_cast<uint32_t>(shard->n_good_cached), B.alleles.p + row0, nr, shard->d_pa.p, reinterpret_cast<LocEntry*>(cells), stride, run,
                     B.counters.p + 2 * slot + 1, stream);
}

void RowGatherer::place_chunk(const uint8_t* cells, uint32_t shard, uint32_t row0) {
    const uint32_t nr = std::min(rows_per_chunk, n_rows - row0);
    const uint64_t n = static_cast<uint64_t>(nr) * goods[shard];
    if (!n) return;
    launch_place_rows(n, reinterpret_cast<const LocEntry*>(cells), stride, static_cast<uint32_t>(goods[shard]), nr,
                      static_cast<uint32_t>(shard * ext_stride), reinterpret_cast<LocCell*>(owner->gather.table.p) + static_cast<size_t>(row0) * ngp,
                      owner->gather.ext.p + static_cast<size_t>(row0) * ngp, owner->gather.unm.p, row0 == 0, ngp, first[shard], stream);
}

void RowGatherer::finish() {
    const uint64_t n = static_cast<uint64_t>(n_rows) * (ngp - n_good);
    if (n)
        launch_pad_rows(n, reinterpret_cast<LocCell*>(owner->gather.table.p), owner->gather.ext.p, owner->gather.unm.p, ngp, n_good, n_rows, stream);
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
    reads->check_device_error(s);
    const uint32_t A = reads->locus->n_alleles;
    for (uint32_t i = 0; i < ploidy; i++)
        if (genotype[i] >= A) fail(LCTY_ERR_INVALID_INPUT, "genotype refers to allele %u >= %u", genotype[i], A);
    // buffers of the batch, grow-only: allocating or freeing here would wait for every stream of the device (the next locus' kernels)
    DevBuf<uint16_t>& d_ids = reads->d_unexpl_ids; DevBuf<unsigned long long>& d_out = reads->d_unexpl_out;
    d_ids.ensure(ploidy); d_ids.upload(genotype, ploidy, s);
    d_out.ensure(1); d_out.zero(s);
    const uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>((reads->n_pairs + 255) / 256, 4096));
    if (reads->n_pairs)
        launch_count_unexplained(blocks, reads->d_status.p, reads->d_unmapped.p, reads->d_matrix.p, reads->n_pairs, A, d_ids.p, ploidy, d_out.p, s);
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
            launch_read_nw(R.V, d_nw.p, s);
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
            launch_assignment_counts(R.V, d_nw.p, d_off.p, d_counts.p, s);
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
        if (!reads || reads->ctx->diag_knob("queue_trace", 0) == 0) return;
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

    // everything before the chains: scores, run_filter, the batch's location table. Issued on the calling thread's stream.
    void pre(bool score) {
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
        ensure_solver_tables(reads);
        mark("head: location table built");
    }
    void chains() { for (uint32_t si = 0; si + 1 < n_stages; si++) stage(si, 0); }
    void head(bool score) { pre(score); chains(); }

    void tail(uint32_t lane) {
        lcty_ctx* ctx = reads->ctx;
        ctx->activate();
        // on the side lane every call of this thread that says "the context's stream" means the side stream (a wider depth table, the error flag)
        std::unique_ptr<StreamScope> on_side;
        if (lane) on_side = std::make_unique<StreamScope>(ctx->side_stream());
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
// last stage has been joined.
// What comes before the chains of locus i + 1 (LocusRun::pre: scores, run_filter, the cut, the location table — 19 + 2.5 ms of a 380-ms
// step at 1 M read pairs x 256 alleles) is issued by a third host thread on the context's fore stream as soon as the last stage of locus
// i - 1 has ended, i.e. beside the greedy chains of locus i, which leave the device's issue slots and a fifth of its LDS free (knob
// "queue_early_head" 0: on the main stream after those chains). Batch i + 1 is then touched while locus i is in its chains — never
// before the tail of i - 1 has ended, so "an entry may appear again, not next to itself" still holds. `batch_of(i + 1)` is called from
// that thread (a fed queue's loader needs the time until then), `done_with` from the caller's: a fed queue's callbacks may run at the
// same time, and position i + 1 is acquired BEFORE position i - 1 is released (three batch objects still carry any length).
template <typename GET, typename DONE>
void run_queue(uint32_t n, GET&& batch_of, DONE&& done_with, uint32_t ploidy, const lcty_stage* stages, uint32_t n_stages,
               const uint64_t* master_seeds, const double* const* priors, lcty_call* out) {
    std::unique_ptr<LocusRun> prev, next;
    std::thread tail_thread, fore_thread;
    int32_t tail_rc = LCTY_OK; std::string tail_msg;
    int32_t fore_rc = LCTY_OK; std::string fore_msg;
    std::shared_future<void> tail_ended;                                     // of the tail that is running (or the last one that ran)
    lcty_ctx* ctx = nullptr;
    uint32_t tail_of = 0;
    auto release_gate = [&] {
        if (!ctx) return;
        { std::lock_guard<std::mutex> lock(ctx->gate.m); if (ctx->gate.target > ctx->gate.epoch) ctx->gate.epoch = ctx->gate.target; }
        ctx->gate.cv.notify_all();
    };
    // positions whose batch has been asked for and not yet given back: on an error every one of them is given back once nothing of the
    // queue is in flight any more (the fore thread acquires position i + 1 while i is in its chains and i - 1 in its tail)
    std::mutex held_mutex;
    std::vector<uint32_t> held;
    auto give_back = [&](uint32_t i) {
        { std::lock_guard<std::mutex> lock(held_mutex); held.erase(std::remove(held.begin(), held.end(), i), held.end()); }
        done_with(i);
    };
    auto join_tail = [&] {
        const bool had = tail_thread.joinable();
        if (had) tail_thread.join();
        prev.reset();
        if (tail_rc != LCTY_OK) { const int32_t rc = tail_rc; tail_rc = LCTY_OK; fail(rc, "%s", tail_msg.c_str()); }
        if (had) give_back(tail_of);
    };
    auto join_fore = [&] {
        if (fore_thread.joinable()) fore_thread.join();
        if (fore_rc != LCTY_OK) { const int32_t rc = fore_rc; fore_rc = LCTY_OK; next.reset(); fail(rc, "%s", fore_msg.c_str()); }
    };
    auto make_run = [&](uint32_t i) {
        auto R = std::make_unique<LocusRun>();
        R->reads = batch_of(i);
        { std::lock_guard<std::mutex> lock(held_mutex); held.push_back(i); }
        R->ploidy = ploidy; R->stages = stages; R->n_stages = n_stages; R->master_seed = master_seeds[i];
        R->priors = priors ? priors[i] : nullptr; R->out = &out[i];
        return R;
    };
    try {
        for (uint32_t i = 0; i < n; i++) {
            std::unique_ptr<LocusRun> R;
            if (fore_thread.joinable()) { join_fore(); R = std::move(next); } // its scores, cut and tables were made beside the chains of locus i - 1
            else { R = make_run(i); R->pre(true); }
            ctx = R->reads->ctx;
            if (i + 1 < n && ctx->knob("queue_early_head", 1) != 0) {
                std::shared_future<void> before = tail_ended;               // the tail of locus i - 1: batch i + 1 may be the batch it works on
                lcty_ctx* c = ctx;
                // (`batch_of(i + 1)` from that thread too, once the tail has ended: a fed queue's loader is filling that batch meanwhile)
                fore_thread = std::thread([&next, &make_run, i, before, c, &fore_rc, &fore_msg] {
                    try {
                        if (before.valid()) before.wait();
                        c->activate();
                        next = make_run(i + 1);
                        if (next->reads->ctx != c) fail(LCTY_ERR_INVALID_INPUT, "the batches of a queue share one context");
                        LocusRun* nx = next.get();
                        StreamScope on_fore(c->fore_stream());
                        nx->pre(true);
                        LCTY_HIP(hipStreamSynchronize(c->fore_stream()));       // the chains of this locus are issued on the main stream
                    }
                    catch (const Error& e) { fore_rc = e.code; fore_msg = e.what(); }
                    catch (const std::exception& e) { fore_rc = LCTY_ERR_RUNTIME; fore_msg = e.what(); }
                });
            }
            try { R->chains(); }
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
            auto ended = std::make_shared<std::promise<void>>();
            tail_ended = ended->get_future().share();
            { std::lock_guard<std::mutex> lock(ctx->gate.m); ctx->gate.tails_started++; }
            lcty_ctx* c = ctx;
            // (a tail that cannot be started has no initialisation to wait for: the count must not stay ahead, or the next stage of this
            // context — this queue's or a later call's — would wait for it for ever)
            struct StartGuard {
                lcty_ctx* c; bool armed = true;
                ~StartGuard() {
                    if (!armed) return;
                    { std::lock_guard<std::mutex> lock(c->gate.m); c->gate.tail_inits = c->gate.tails_started; }
                    c->gate.cv.notify_all();
                }
            } start_guard{c};
            tail_thread = std::thread([run, ended, c, &tail_rc, &tail_msg] {
                try { run->tail(1); }
                catch (const Error& e) { tail_rc = e.code; tail_msg = e.what(); }
                catch (const std::exception& e) { tail_rc = LCTY_ERR_RUNTIME; tail_msg = e.what(); }
                // (a tail that ended without an initialisation of its own — an error, a skipped stage — must not keep the next locus waiting)
                { std::lock_guard<std::mutex> lock(c->gate.m); if (c->gate.tail_inits < c->gate.tails_started) c->gate.tail_inits = c->gate.tails_started; }
                c->gate.cv.notify_all();
                ended->set_value();
            });
            start_guard.armed = false;
        }
        join_tail();
    } catch (...) {
        release_gate();
        if (fore_thread.joinable()) fore_thread.join();
        if (tail_thread.joinable()) tail_thread.join();
        // Nothing of the queue may still be running when the caller gets its batches back (it may reset or destroy them): what the
        // three threads issued is waited for, then every position that was acquired and not released is released — the one whose
        // tail was joined here, the one in its chains, and the one the fore thread had taken ahead.
        if (ctx) {
            (void)hipStreamSynchronize(ctx->stream.main);
            if (ctx->side) (void)hipStreamSynchronize(ctx->side);
            if (ctx->fore) (void)hipStreamSynchronize(ctx->fore);
        }
        std::vector<uint32_t> left;
        { std::lock_guard<std::mutex> lock(held_mutex); left.swap(held); }
        for (uint32_t i : left) {
            try { done_with(i); } catch (...) {}                              // the first error is the one reported
        }
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

// The same queue with the batches handed over one at a time: `acquire(user, i)` returns the batch of position i — filled by then,
// e.g. by a host thread that appends the chunks of locus i while locus i - 1 is being solved (the appends have a stream of their
// own) —, `release(user, i)` is called when the last stage of position i has finished and nothing of the batch is in use any more
// (lcty_reads_reset may then bind it to another locus). WHICH THREAD CALLS WHAT (round 5 on): acquire(0) comes from the caller's
// thread; acquire(i > 0) from a thread of the library, as soon as the last stage of position i - 2 has ended, i.e. while position
// i - 1 is in its chains — so acquire(i) can run AT THE SAME TIME as release(i - 2) and comes BEFORE release(i - 1); release(i)
// precedes acquire(i + 3): three batch objects carry a queue of any length. Callbacks that share state guard it (or set the knob
// "queue_early_head" to 0: every callback then comes from the caller's thread, acquire(i) after release(i - 2)). On an error every
// position that was acquired and not yet released is released before the call returns, with nothing of the queue left in flight.
// The loop of `locityper genotype` over its loci (genotype.rs:1331-1351) with the loading of locus i + 1 next to the solving of
// locus i.
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
