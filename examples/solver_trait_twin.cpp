// The Rust shim's `impl Solver for HipSolver` (shim/src/solvers/hip.rs) once more in C++, statement for statement, so that the call
// sequence of the shim is COMPILED against include/locityper_hip.h and RUN on the device (no rustc in the image this library is built
// in): flatten a GenotypeAlignments, ask how deep a window can get, evaluate the window distributions as rows of ln_pmf, seed a
// stream from one draw of the caller's generator, lcty_solve_given_tables — from several worker threads at once, as
// solve_multi_thread calls `stage.solver.solve(&gt_alns, rng)` (src/solvers/solve.rs:1010-1017, 1124-1125).
//
//   solver_trait_twin <dump> <out> <threads>
// <dump>: objects written by tests/test_gpu_example.py from oracle-built GenotypeAlignments — per object the arrays the Rust side gets
// from `possible_read_alns`, `depth_distr`, `contributions`, `gt_windows` (model/assgn.rs:86-131), each window's distribution as
// (distribution id, weight), the distributions as rows of LinearCache::ln_pmf, the solver and the u64 the caller's generator draws.
// <out>: per object read_assgn (u16 per read) and the likelihood, for the test to hold against the oracle's chain.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "locityper_hip.h"

namespace {

struct Object {                       // what `&GenotypeAlignments` gives the shim
    std::vector<std::vector<double>> ln_prob;               // possible_read_alns(rp)[t].ln_prob()
    std::vector<std::vector<uint32_t>> windows;             // ... .windows(), two per location
    std::vector<uint32_t> distr_of_window;                  // depth_distr(w).inner(): which cached distribution (0xFFFFFFFF: TRIVIAL)
    std::vector<double> weight;                             // depth_distr(w).weight()
    std::vector<uint32_t> wshifts;
    double depth_contrib = 0, aln_contrib = 0;
    lcty_solver solver{};
    uint64_t rng_draw = 0;                                  // rng.next_u64() of the caller's XoshiroRng
    // results
    std::vector<uint16_t> read_assgn; double likelihood = 0; int32_t status = -1; std::string error;
};

template <typename T>
bool rd(FILE* f, T* p, size_t n) { return fread(p, sizeof(T), n, f) == n; }

// HipSolver::solve_nontrivial (shim/src/solvers/hip.rs), line for line
void solve_nontrivial(lcty_ctx* ctx, const std::vector<std::vector<double>>& distr_rows, Object& o) {
    const size_t n_reads = o.ln_prob.size();
    std::vector<uint64_t> read_ixs; std::vector<double> ln_prob; std::vector<uint32_t> windows;
    read_ixs.push_back(0);
    for (size_t rp = 0; rp < n_reads; rp++) {
        for (size_t t = 0; t < o.ln_prob[rp].size(); t++) {
            ln_prob.push_back(o.ln_prob[rp][t]);
            windows.push_back(o.windows[rp][2 * t]); windows.push_back(o.windows[rp][2 * t + 1]);
        }
        read_ixs.push_back(ln_prob.size());
    }
    const size_t n_windows = o.weight.size();
    std::vector<uint32_t> distrs;                                           // the distinct distributions this object points to
    std::vector<uint8_t> window_row(n_windows, 0);
    std::vector<double> window_weight(n_windows, 0.0);
    for (size_t w = 0; w < n_windows; w++) {
        if (o.distr_of_window[w] == 0xFFFFFFFFu) continue;                  // WindowDistr::TRIVIAL
        size_t row = 0;
        while (row < distrs.size() && distrs[row] != o.distr_of_window[w]) row++;
        if (row == distrs.size()) distrs.push_back(o.distr_of_window[w]);
        window_row[w] = static_cast<uint8_t>(row);
        window_weight[w] = o.weight[w];
    }
    lcty_gt_alns_view view{};
    view.n_reads = n_reads; view.read_ixs = read_ixs.data(); view.ln_prob = ln_prob.data(); view.windows = windows.data();
    view.n_windows = static_cast<uint32_t>(n_windows); view.n_contigs = static_cast<uint32_t>(o.wshifts.size() - 1);
    view.window_gc = window_row.data(); view.window_weight = window_weight.data(); view.wshifts = o.wshifts.data();
    view.depth_contrib = o.depth_contrib; view.aln_contrib = o.aln_contrib;

    uint32_t deepest = 0;
    if ((o.status = lcty_gt_alns_deepest(&view, &deepest)) != LCTY_OK) { o.error = lcty_last_error(); return; }
    const size_t width = (static_cast<size_t>(deepest) + 1 + 255) / 256 * 256;
    const size_t n_rows = distrs.empty() ? 1 : distrs.size();
    std::vector<double> values(n_rows * width, 0.0);
    for (size_t i = 0; i < distrs.size(); i++)
        for (size_t k = 0; k < width; k++) values[i * width + k] = distr_rows[distrs[i]][k];      // LinearCache::ln_pmf(k)
    lcty_depth_tables tables{static_cast<uint32_t>(n_rows), static_cast<uint32_t>(width), values.data(), 0};

    uint64_t state[4];
    if ((o.status = lcty_rng_seed_from_u64(o.rng_draw, state)) != LCTY_OK) { o.error = lcty_last_error(); return; }
    o.read_assgn.assign(n_reads, 0);
    o.status = lcty_solve_given_tables(ctx, &view, &tables, &o.solver, state, o.read_assgn.data(), nullptr, &o.likelihood);
    if (o.status != LCTY_OK) o.error = lcty_last_error();
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s <dump> <out> <threads>\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    uint32_t n_distr = 0, row_width = 0, n_objects = 0;
    if (!rd(f, &n_distr, 1) || !rd(f, &row_width, 1) || !rd(f, &n_objects, 1)) return 2;
    std::vector<std::vector<double>> distr_rows(n_distr, std::vector<double>(row_width));
    for (auto& r : distr_rows) if (!rd(f, r.data(), row_width)) return 2;
    std::vector<Object> objects(n_objects);
    for (Object& o : objects) {
        uint64_t n_reads = 0; uint32_t n_windows = 0, n_contigs = 0;
        if (!rd(f, &n_reads, 1) || !rd(f, &n_windows, 1) || !rd(f, &n_contigs, 1)) return 2;
        std::vector<uint64_t> ixs(n_reads + 1);
        if (!rd(f, ixs.data(), n_reads + 1)) return 2;
        o.ln_prob.resize(n_reads); o.windows.resize(n_reads);
        for (uint64_t rp = 0; rp < n_reads; rp++) {
            const size_t n = ixs[rp + 1] - ixs[rp];
            o.ln_prob[rp].resize(n); o.windows[rp].resize(2 * n);
        }
        for (uint64_t rp = 0; rp < n_reads; rp++) if (!rd(f, o.ln_prob[rp].data(), o.ln_prob[rp].size())) return 2;
        for (uint64_t rp = 0; rp < n_reads; rp++) if (!rd(f, o.windows[rp].data(), o.windows[rp].size())) return 2;
        o.distr_of_window.resize(n_windows); o.weight.resize(n_windows); o.wshifts.resize(n_contigs + 1);
        if (!rd(f, o.distr_of_window.data(), n_windows) || !rd(f, o.weight.data(), n_windows) || !rd(f, o.wshifts.data(), n_contigs + 1)) return 2;
        if (!rd(f, &o.depth_contrib, 1) || !rd(f, &o.aln_contrib, 1) || !rd(f, &o.solver, 1) || !rd(f, &o.rng_draw, 1)) return 2;
    }
    fclose(f);

    lcty_ctx* ctx = nullptr;                                                  // HipCtx::global() (shim/src/hip/mod.rs)
    if (lcty_ctx_create(0, &ctx) != LCTY_OK) { fprintf(stderr, "%s\n", lcty_last_error()); return 1; }
    // solve_multi_thread: the worker threads share the solver (`&self`) and take (genotype, attempt) units as they come
    const int n_threads = std::max(1, atoi(argv[3]));
    std::atomic<size_t> next{0};
    std::vector<std::thread> workers;
    for (int t = 0; t < n_threads; t++)
        workers.emplace_back([&] { for (size_t i; (i = next.fetch_add(1)) < objects.size();) solve_nontrivial(ctx, distr_rows, objects[i]); });
    for (auto& w : workers) w.join();

    FILE* g = fopen(argv[2], "wb");
    if (!g) return 2;
    int failed = 0;
    for (const Object& o : objects) {
        if (o.status != LCTY_OK) { fprintf(stderr, "status %d: %s\n", o.status, o.error.c_str()); failed++; continue; }
        const uint64_t n = o.read_assgn.size();
        fwrite(&n, sizeof(n), 1, g); fwrite(o.read_assgn.data(), sizeof(uint16_t), n, g); fwrite(&o.likelihood, sizeof(double), 1, g);
    }
    fclose(g);
    lcty_ctx_destroy(ctx);
    printf("solved %zu objects on %d threads, %d failed\n", objects.size(), n_threads, failed);
    return failed ? 1 : 0;
}
