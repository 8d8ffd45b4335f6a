// lcty_exact.hpp — the exact solver of a chain's model (the place of HighsSolver / GurobiSolver,
// src/solvers/highs.rs:38-134, gurobi.rs:15-83) as plain host C++: no device types, no HIP calls. lcty_solve_host.hip builds the
// model from what solve_init_kernel left on the device, hands models to a pool of host threads (the reference runs one model per
// worker thread, solve.rs:1052-1062) and writes the assignments back into the chains' records.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace lcty {
namespace exact {

// one possible location of a non-trivial read: its ln-probability and its two windows after apply_tweak
struct Loc { double lp; uint32_t wa, wb; };

// The integer programme of highs.rs:38-100 in the terms the search uses: read i has the locations locs[first[i] .. first[i + 1]) in
// the order of extend_read_gt_alns (best first); window w has weight ww[w] (0 = WindowDistr::TRIVIAL), GC bin gcb[w] and, with every
// non-trivial read at its location 0, depth depth0[w]; aln0 = the alignment likelihood of that start (trivial reads included).
struct Model {
    uint32_t n = 0, tw = 0;
    std::vector<uint32_t> first;
    std::vector<Loc> locs;
    std::vector<double> ww;
    std::vector<uint8_t> gcb;
    std::vector<uint32_t> depth0;
    double aln0 = 0.0;
    std::vector<uint32_t> allele_first_w;       // first window of every allele of the genotype, and one past the last
    double aln_contrib = 1.0, depth_contrib = 1.0;
    uint64_t node_limit = 20ull * 1000 * 1000;
    double rel_gap = 0.0;                        // HiGHS' mip_rel_gap; 0 = a proof of optimality
    uint32_t chain = 0;                          // for the trace lines only
    int trace = 0;
    std::string dump_path;                       // developer dump of the model as the search sees it after the fixing rounds (text); "" = none
    uint32_t gc_bins = 0;                        // rows of the depth table (for the dump)
};

struct Result {
    std::vector<uint8_t> assign;                 // location of every non-trivial read
    double value = 0.0, depth_lik = 0.0, aln_lik = 0.0;
    uint64_t nodes = 0;
    bool out_of_nodes = false;
    uint32_t n_free = 0;
};

// deepest table entry the search can ask for (+ 1): the caller widens the depth table to at least this before `solve`
uint64_t depth_needed(const Model& m);

// lut = the depth table [GC bins][ld] (WindowDistr::ln_prob = weight * lut[gc][depth], distr_cache.rs:34-39)
void solve(const Model& m, const double* lut, uint32_t ld, Result& out);

}  // namespace exact
}  // namespace lcty
