// lcty_api.hip — context, parameters, prefilter / truncate entry points, timing hooks.
#include <algorithm>
#include <cmath>
#include <memory>
#include <numeric>

#include "lcty_objects.hpp"

using namespace lcty;

namespace lcty {
static thread_local std::string g_last_error;
thread_local hipStream_t tl_stream = nullptr;
void set_last_error(const std::string& msg) { g_last_error = msg; }
}  // namespace lcty

hipEvent_t lcty_ctx::get_event() {
    hipEvent_t e;
    if (!event_pool.empty()) { e = event_pool.back(); event_pool.pop_back(); return e; }
    LCTY_HIP(hipEventCreate(&e));
    return e;
}

// the oldest `count` event pairs of a timer are waited for and added up (bounds what a long timed run keeps alive)
void lcty_ctx::fold_oldest(lcty::KernelTimer& t, size_t count) {
    count = std::min(count, t.pending.size());
    for (size_t i = 0; i < count; i++) {
        float ms = 0.f;
        LCTY_HIP(hipEventSynchronize(t.pending[i].second));
        LCTY_HIP(hipEventElapsedTime(&ms, t.pending[i].first, t.pending[i].second));
        t.total_ms += ms; t.launches++;
        event_pool.push_back(t.pending[i].first); event_pool.push_back(t.pending[i].second);
    }
    t.pending.erase(t.pending.begin(), t.pending.begin() + static_cast<std::ptrdiff_t>(count));
}

extern "C" {

// Limits of the retry / batching machinery, settable per context so that tests can reach those paths with small inputs
// (a value < 0 restores the default). Nothing here changes a result; an unknown name is an error.
int32_t lcty_ctx_set_knob(lcty_ctx* ctx, const char* name, int64_t value) {
    return guarded([&] {
        if (!ctx || !name) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        static const char* const known[] = {"transfer_levels", "transfer_scratch_mb", "transfer_waves", "transfer_cap_new", "transfer_arena",
                                            "depth_table_start", "solve_budget_mb", "solve_chains_per_wave", "solve_extra_start", "solve_lds_weights",
                                            "anneal_lds_weights", "contig_info_slide", "gather_chunk_mb", "prefilter_gram", "prefilter_gram_cols",
                                            "prefilter_gram_levels", "comm_fail_at", "score_lean", "arena_cap_pct", "exact_threads", "host_threads",
                                            "score_lean_keep", "score_lean_two", "queue_early_head",
#ifdef LCTY_DIAG
                                            // the developer build (make DIAG=1): traces, in-kernel timing, kernel forms under measurement
                                            "solve_stats", "queue_trace", "map_trace", "exact_trace", "solve_greedy_form", "solve_anneal_timing", "score_timing",
                                            "solve_init_tiles", "transfer_phases",
#endif
                                            nullptr};
        bool ok = false;
        for (const char* const* k = known; *k; k++) ok |= strcmp(*k, name) == 0;
        if (!ok) fail(LCTY_ERR_INVALID_INPUT, "unknown knob '%s'", name);
        if (value < 0) ctx->knobs.erase(name); else ctx->knobs[name] = value;
    });
}

// Files a developer asks the library to leave behind (nothing is ever read from the environment): "exact_dump" = the model of the
// first chain of an exact-solver stage as text (scripts/exact_probe.py --dump); NULL or "" switches it off.
int32_t lcty_ctx_set_path(lcty_ctx* ctx, const char* name, const char* path) {
    return guarded([&] {
        if (!ctx || !name) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (strcmp(name, "exact_dump") != 0) fail(LCTY_ERR_INVALID_INPUT, "unknown path '%s'", name);
        ctx->exact_dump_path = path ? path : "";
    });
}

const char* lcty_last_error(void) { return g_last_error.c_str(); }
const char* lcty_version(void) { return "locityper_hip 0.1.0 (gfx950)"; }

int32_t lcty_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int32_t lcty_ctx_create(int32_t device_id, lcty_ctx** out) {
    return guarded([&] {
        if (!out) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
            (void)hipGetLastError();
            fail(LCTY_ERR_RUNTIME, "no HIP device is visible: liblocityper_hip has no CPU fallback");
        }
        if (device_id < 0 || device_id >= n) fail(LCTY_ERR_INVALID_INPUT, "device %d out of range (0..%d)", device_id, n - 1);
        auto c = std::unique_ptr<lcty_ctx>(new lcty_ctx());
        c->device = device_id;
        LCTY_HIP(hipSetDevice(device_id));
        LCTY_HIP(hipGetDeviceProperties(&c->props, device_id));
        LCTY_HIP(hipStreamCreateWithFlags(&c->stream.main, hipStreamNonBlocking));
        *out = c.release();
    });
}

void lcty_ctx_destroy(lcty_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& t : ctx->timers) for (auto& pr : t.pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->gate.ev) (void)hipEventDestroy(ctx->gate.ev);
    if (ctx->gate.init_ev) (void)hipEventDestroy(ctx->gate.init_ev);
    if (ctx->side) { (void)hipStreamSynchronize(ctx->side); (void)hipStreamDestroy(ctx->side); }
    if (ctx->fore) { (void)hipStreamSynchronize(ctx->fore); (void)hipStreamDestroy(ctx->fore); }
    for (auto& slot : ctx->given_slots) if (slot->stream) { (void)hipStreamSynchronize(slot->stream); (void)hipStreamDestroy(slot->stream); }
    (void)hipStreamDestroy(ctx->stream.main);
    delete ctx;
}

int32_t lcty_ctx_synchronize(lcty_ctx* ctx) {
    return guarded([&] {
        if (!ctx) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        ctx->activate();
        LCTY_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->side) LCTY_HIP(hipStreamSynchronize(ctx->side));
    });
}

// releases the solver workspaces the context keeps between stages and loci (they are rebuilt by the next stage)
int32_t lcty_host_alloc(lcty_ctx* ctx, uint64_t bytes, void** out) {
    return guarded([&] {
        if (!ctx || !out) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        ctx->activate();
        *out = nullptr;
        LCTY_HIP(hipHostMalloc(out, std::max<uint64_t>(bytes, 1), hipHostMallocDefault));
    });
}

void lcty_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int32_t lcty_ctx_trim(lcty_ctx* ctx) {
    return guarded([&] {
        if (!ctx) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        ctx->activate();
        LCTY_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->side) LCTY_HIP(hipStreamSynchronize(ctx->side));
        for (auto& w : ctx->solve_ws) w.release_all();
        {
            std::lock_guard<std::mutex> g(ctx->given_mutex);
            for (auto& slot : ctx->given_slots) {
                if (slot->busy) continue;                                     // a call of another thread is using it
                slot->ws.release_all(); slot->lut.release(); slot->lut_depth = 0; slot->lut_of = 0; slot->lut_is_given = false;
                slot->read_ixs.release(); slot->lp.release(); slot->weight.release(); slot->win.release(); slot->gc.release(); slot->assgn.release();
            }
        }
        ctx->release_transfer_scratch();
    });
}

// model::Params::default — src/model/mod.rs:108-135
void lcty_params_default(lcty_params* p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->boundary_size = 200;
    p->tweak = -1;
    p->lik_skew = 0.85;
    p->prob_diff = std::numeric_limits<double>::quiet_NaN();
    p->unmapped_penalty = std::numeric_limits<double>::quiet_NaN();
    p->poor_compl = 0.5;
    p->poor_compl_edit = 0.7;
    p->compl_weight_bp = 0.5; p->compl_weight_pow = 4.0;
    p->kmers_weight_bp = 0.2; p->kmers_weight_pow = 4.0;
    p->min_weight = 0.001;
    p->filt_diff = 100.0 * math::LN10;     // Ln::from_log10(100.0)
    p->prob_thresh = -4.0 * math::LN10;
    const double cn[5] = {0.3, 2.0, 3.0, 4.0, 5.0};
    for (int i = 0; i < 5; i++) p->alt_cn[i] = cn[i];
    p->n_alt_cn = 5;
    p->kmer_soft_thresh = 5;
    p->kmer_hard_thresh = 1;
    p->complexity_k = 5;
    p->threads = 8;                         // src/command/genotype.rs:127
}

// Params::set_tweak_size (model/mod.rs:179-197) + command/genotype.rs:1291-1296 + Params::validate (144-177)
int32_t lcty_params_resolve(lcty_params* p, const lcty_bg* bg) {
    return guarded([&] {
        if (!p || !bg) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (p->boundary_size == 0) fail(LCTY_ERR_INVALID_INPUT, "Boundary size (0) cannot be zero.");
        if (!(p->lik_skew >= -1.0 + 1e-10 && p->lik_skew <= 1.0 - 1e-10))
            fail(LCTY_ERR_INVALID_INPUT, "Likelihood skew (%g) must be within (-1, 1)", p->lik_skew);
        if (!(0.0 <= p->min_weight && p->min_weight <= 0.5))
            fail(LCTY_ERR_INVALID_INPUT, "Minimal weight (%g) must be within [0, 0.5].", p->min_weight);
        if (p->tweak < 0) {
            uint32_t t = static_cast<uint32_t>(std::round(static_cast<double>(bg->window) * 0.5));
            t = std::min(t, 200u);
            t = std::min(t, p->boundary_size - 1);
            p->tweak = static_cast<int32_t>(t);
        }
        if (static_cast<uint32_t>(p->tweak) >= p->boundary_size)
            fail(LCTY_ERR_INVALID_INPUT, "Boundary size (%u) must be greater than tweak size (%d).", p->boundary_size, p->tweak);
        if (static_cast<uint32_t>(p->tweak) > 65535u / 2 - 1)
            fail(LCTY_ERR_INVALID_INPUT, "Tweaking size (%d) is too large (max = %u)", p->tweak, 65535u / 2 - 1);
        if (std::isnan(p->unmapped_penalty))
            p->unmapped_penalty = (bg->technology == LCTY_TECH_ILLUMINA ? -10.0 : -100.0) * math::LN10;
        if (std::isnan(p->prob_diff)) p->prob_diff = std::fabs(p->unmapped_penalty) + math::LN10;
        p->prob_diff = std::fabs(p->prob_diff);
        if (!(p->filt_diff >= 0.0)) fail(LCTY_ERR_INVALID_INPUT, "Filtering likelihood difference must be non-negative");
        if (!(p->prob_thresh < 0.0)) fail(LCTY_ERR_INVALID_INPUT, "Probability threshold must be negative");
    });
}

uint64_t lcty_count_genotypes(uint32_t n_alleles, uint32_t ploidy) { return count_genotypes(n_alleles, ploidy); }

// gen_combinations_with_repl — src/ext/vec.rs:298-339 (iterative odometer instead of the recursion)
int32_t lcty_generate_genotypes(uint32_t n_alleles, uint32_t ploidy, uint16_t* out, uint64_t cap) {
    return guarded([&] {
        if (!out) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (n_alleles == 0 || ploidy == 0) return;
        if (n_alleles > 65535) fail(LCTY_ERR_INVALID_INPUT, "too many alleles");
        const uint64_t total = count_genotypes(n_alleles, ploidy);
        if (cap < total) fail(LCTY_ERR_INVALID_INPUT, "output capacity %llu < %llu genotypes", (unsigned long long)cap, (unsigned long long)total);
        std::vector<uint32_t> cur(ploidy, 0);
        for (uint64_t g = 0; g < total; g++) {
            for (uint32_t t = 0; t < ploidy; t++) out[g * ploidy + t] = static_cast<uint16_t>(cur[t]);
            // next non-decreasing tuple in lexicographic order
            int32_t t = static_cast<int32_t>(ploidy) - 1;
            while (t >= 0 && cur[t] == n_alleles - 1) t--;
            if (t < 0) break;
            const uint32_t v = cur[t] + 1;
            for (uint32_t u = t; u < ploidy; u++) cur[u] = v;
        }
    });
}

static void require_scored(lcty_reads* reads) {
    if (!reads) fail(LCTY_ERR_INVALID_INPUT, "null argument");
    if (!reads->scored) fail(LCTY_ERR_INVALID_INPUT, "lcty_score_reads has not been called on this batch");
    reads->ctx->activate();
}

int32_t lcty_prefilter_async(lcty_reads* reads, uint32_t ploidy) {
    return guarded([&] {
        require_scored(reads);
        if (ploidy != 2) fail(LCTY_ERR_UNSUPPORTED, "lcty_prefilter_async handles ploidy 2; use lcty_prefilter for other ploidies");
        launch_prefilter_diploid(reads);
    });
}

int32_t lcty_prefilter_scores(lcty_reads* reads, double* scores, uint64_t n) {
    return guarded([&] {
        require_scored(reads);
        if (!scores) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (n != reads->n_scores) fail(LCTY_ERR_INVALID_INPUT, "expected %llu scores", (unsigned long long)reads->n_scores);
        reads->check_device_error();
        reads->d_scores.download(scores, n, reads->ctx->stream);
        LCTY_HIP(hipStreamSynchronize(reads->ctx->stream));
    });
}

int32_t lcty_prefilter(lcty_reads* reads, const uint16_t* genotypes, uint64_t n_genotypes, uint32_t ploidy,
                       const double* priors, double* scores) {
    return guarded([&] {
        require_scored(reads);
        if (!scores) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (ploidy == 0) fail(LCTY_ERR_INVALID_INPUT, "ploidy must be positive");
        reads->check_device_error();
        hipStream_t s = reads->ctx->stream;
        const uint32_t A = reads->locus->n_alleles;
        if (!genotypes && ploidy == 2) {
            const uint64_t G = count_genotypes(A, 2);
            if (n_genotypes != G) fail(LCTY_ERR_INVALID_INPUT, "expected %llu genotypes, got %llu", (unsigned long long)G, (unsigned long long)n_genotypes);
            launch_prefilter_diploid(reads);
            reads->d_scores.download(scores, G, s);
            LCTY_HIP(hipStreamSynchronize(s));
            if (priors) for (uint64_t g = 0; g < G; g++) scores[g] = priors[g] + scores[g];    // prior + sum (solve.rs:114)
            return;
        }
        std::vector<uint16_t> all;
        if (!genotypes) {
            const uint64_t G = count_genotypes(A, ploidy);
            if (n_genotypes != G) fail(LCTY_ERR_INVALID_INPUT, "expected %llu genotypes, got %llu", (unsigned long long)G, (unsigned long long)n_genotypes);
            all.resize(G * ploidy);
            if (lcty_generate_genotypes(A, ploidy, all.data(), G) != LCTY_OK) fail(LCTY_ERR_RUNTIME, "%s", lcty_last_error());
            genotypes = all.data();
        }
        for (uint64_t i = 0; i < n_genotypes * ploidy; i++)
            if (genotypes[i] >= A) fail(LCTY_ERR_INVALID_INPUT, "genotype refers to allele %u >= %u", genotypes[i], A);
        DevBuf<uint16_t> d_gt; DevBuf<double> d_pr, d_sc;
        d_gt.alloc(n_genotypes * ploidy); d_gt.upload(genotypes, n_genotypes * ploidy, s);
        if (priors) { d_pr.alloc(n_genotypes); d_pr.upload(priors, n_genotypes, s); }
        d_sc.alloc(n_genotypes);
        launch_prefilter_generic(reads, d_gt.p, n_genotypes, ploidy, priors ? d_pr.p : nullptr, d_sc.p);
        d_sc.download(scores, n_genotypes, s);
        LCTY_HIP(hipStreamSynchronize(s));
    });
}

// KmerCounts::load — src/seq/counts.rs:127-150: the first KmerCounts block of `kmers.bin` (after the caller has taken the
// brotli / lz4 layer off, ext/sys.rs:41-76): u8 k, u8 counter bytes, varint n_contigs, then per contig varint n_kmers and n_kmers
// varints, clamped to min(u16::MAX, 2^(8 bytes) - 1). The varints are those of the `varint-rs` crate (^2.2, not in the tree):
// little-endian groups of 7 bits, the high bit of a byte says another one follows. The first block holds the off-target
// counts (command/add.rs:647-650) — the ones lcty_locus_create takes; the second block (all counts) is left unread, as upstream.
int32_t lcty_kmer_counts_parse(const uint8_t* buf, uint64_t len, uint32_t* k_out, uint32_t* n_contigs_out, uint64_t* cnt_off,
                               uint64_t cap_contigs, uint16_t* counts, uint64_t cap_counts, uint64_t* consumed) {
    return guarded([&] {
        if (!buf || !k_out || !n_contigs_out) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        uint64_t at = 0;
        auto byte = [&]() -> uint8_t {
            if (at >= len) fail(LCTY_ERR_INVALID_DATA, "k-mer counts: unexpected end of data at byte %llu", static_cast<unsigned long long>(at));
            return buf[at++];
        };
        auto varint = [&](uint32_t max_bytes) -> uint64_t {
            uint64_t v = 0;
            for (uint32_t i = 0; i < max_bytes; i++) {
                const uint8_t b = byte();
                v |= static_cast<uint64_t>(b & 0x7Fu) << (7 * i);
                if (!(b & 0x80u)) return v;
            }
            fail(LCTY_ERR_INVALID_DATA, "k-mer counts: a varint of more than %u bytes at byte %llu", max_bytes, static_cast<unsigned long long>(at));
            return 0;
        };
        const uint32_t k = byte();
        const uint32_t byte_len = byte();
        if (byte_len > 8) fail(LCTY_ERR_INVALID_DATA, "k-mer counts: counter length of %u bytes", byte_len);            // assert!, counts.rs:133
        const uint64_t max_value = std::min<uint64_t>(0xFFFFu, byte_len == 8 ? ~0ull : (1ull << (byte_len * 8)) - 1);
        const uint64_t n_contigs = varint(5);
        *k_out = k; *n_contigs_out = static_cast<uint32_t>(n_contigs);
        const bool store = cnt_off && counts;
        if (store && cap_contigs < n_contigs) fail(LCTY_ERR_INVALID_INPUT, "k-mer counts: room for %llu contigs, the file has %llu",
                                                   static_cast<unsigned long long>(cap_contigs), static_cast<unsigned long long>(n_contigs));
        uint64_t total = 0;
        if (cnt_off && cap_contigs >= n_contigs) cnt_off[0] = 0;
        for (uint64_t c = 0; c < n_contigs; c++) {
            const uint64_t n_kmers = varint(5);
            if (store && total + n_kmers > cap_counts) fail(LCTY_ERR_INVALID_INPUT, "k-mer counts: room for %llu values is not enough",
                                                            static_cast<unsigned long long>(cap_counts));
            for (uint64_t i = 0; i < n_kmers; i++) {
                const uint64_t v = varint(10);
                if (store) counts[total + i] = static_cast<uint16_t>(std::min(v, max_value));
            }
            total += n_kmers;
            if (cnt_off && cap_contigs >= n_contigs) cnt_off[c + 1] = total;
        }
        if (consumed) *consumed = at;
    });
}

// truncate_ixs — src/solvers/solve.rs:52-84; ties ordered by index ascending
int32_t lcty_truncate(const double* scores, uint64_t* ixs, uint64_t n, double filt_diff, uint64_t min_size,
                      uint64_t threads, uint64_t* n_keep) {
    return guarded([&] {
        if (!scores || !ixs || !n_keep) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (n == 0) fail(LCTY_ERR_INVALID_INPUT, "no genotypes to filter");
        // The reference orders with f64::total_cmp (solve.rs:60) and can never see a NaN here: matrix entries are finite or -inf and
        // priors are checked to be finite when they are read (genotype.rs:1113-1117). `>` is a strict weak order on everything but
        // NaN, so a NaN (a caller's own priors) is refused instead of handed to std::sort / nth_element.
        for (uint64_t t = 0; t < n; t++)
            if (std::isnan(scores[ixs[t]])) fail(LCTY_ERR_INVALID_INPUT, "score of genotype %llu is NaN", static_cast<unsigned long long>(ixs[t]));
        // truncate_ixs (solve.rs:52-84) sorts all indices and keeps a prefix; only the prefix is returned here in sorted order
        // (score descending, ties by index), found by selection: O(n + kept log kept) instead of a sort of 8.4 M indices at
        // 4 096 alleles. What follows the kept prefix in `ixs` is unspecified.
        auto before = [&](uint64_t i, uint64_t j) {
            if (scores[i] != scores[j]) return scores[i] > scores[j];
            return i < j;
        };
        double best = scores[ixs[0]], worst = best;
        for (uint64_t t = 1; t < n; t++) { const double v = scores[ixs[t]]; best = v > best ? v : best; worst = v < worst ? v : worst; }
        double thresh = best - filt_diff;
        if (min_size >= n || worst >= thresh) { std::sort(ixs, ixs + n, before); *n_keep = n; return; }
        auto count_ge = [&](double t) {
            uint64_t c = 0;
            for (uint64_t q = 0; q < n; q++) c += scores[ixs[q]] >= t;
            return c;
        };
        uint64_t m = count_ge(thresh);
        if (m < min_size) {
            std::nth_element(ixs, ixs + (min_size - 1), ixs + n, before);
            thresh = scores[ixs[min_size - 1]];
            m = count_ge(thresh);                       // everything tied with the min_size-th score stays (partition_point)
        }
        const uint64_t at_least = m;                    // the indices with score >= thresh
        m = std::min(std::max(m, threads), n);
        if (m == at_least) std::partition(ixs, ixs + n, [&](uint64_t i) { return scores[i] >= thresh; });   // one pass, no selection
        else std::nth_element(ixs, ixs + m, ixs + n, before);                                             // raised to `threads`
        std::sort(ixs, ixs + m, before);
        *n_keep = m;
    });
}

int32_t lcty_timing_reset(lcty_ctx* ctx) {
    return guarded([&] {
        if (!ctx) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        ctx->activate();
        LCTY_HIP(hipStreamSynchronize(ctx->stream));
        ctx->timing_on = true;                          // timing is opt-in: launches before the first reset are not timed
        for (auto& t : ctx->timers) {
            for (auto& pr : t.pending) { ctx->event_pool.push_back(pr.first); ctx->event_pool.push_back(pr.second); }
            t.pending.clear(); t.launches = 0; t.total_ms = 0.0;
        }
    });
}

int32_t lcty_timing_get(lcty_ctx* ctx, int32_t kernel, uint64_t* launches, double* total_ms) {
    return guarded([&] {
        if (!ctx || kernel < 0 || kernel >= LCTY_K_COUNT) fail(LCTY_ERR_INVALID_INPUT, "bad argument");
        ctx->activate();
        LCTY_HIP(hipStreamSynchronize(ctx->stream));
        auto& t = ctx->timers[kernel];
        for (auto& pr : t.pending) {
            float ms = 0.f;
            LCTY_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
            t.total_ms += ms; t.launches++;
            ctx->event_pool.push_back(pr.first); ctx->event_pool.push_back(pr.second);
        }
        t.pending.clear();
        if (launches) *launches = t.launches;
        if (total_ms) *total_ms = t.total_ms;
    });
}

// count_to_prob (src/model/bam.rs:56-67)
int32_t lcty_counts_to_posteriors(const uint16_t* counts, uint64_t n, uint16_t attempts, float* prob, uint8_t* mapq) {
    return guarded([&] {
        if ((n && !counts) || !prob || !mapq) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        for (uint64_t i = 0; i < n; i++) {
            const uint16_t c = counts[i];
            if (c == 0) { prob[i] = 0.0f; mapq[i] = 0; }
            else if (c == attempts) { prob[i] = 1.0f; mapq[i] = 60; }
            else {
                if (c > attempts) fail(LCTY_ERR_INVALID_INPUT, "count %u of %u attempts", c, attempts);       // assert!(count < attempts)
                const float p = static_cast<float>(c) / static_cast<float>(attempts);
                prob[i] = p;
                mapq[i] = static_cast<uint8_t>(std::fmin(std::round(-10.0f * std::log10(1.0f - p)), 60.0f));
            }
        }
    });
}

}  // extern "C"
