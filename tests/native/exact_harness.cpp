// TEST INFRASTRUCTURE. The library's exact solver (locityper_amd/csrc/lcty_exact.cpp: plain host C++, no device code) behind a flat
// entry point, so that a CPU test can hand it a model built from the oracle's GenotypeAlignments and hold its answer against HiGHS on
// the reference's integer programme (tests/pyref_highs.py). Built by tests/exact_harness.py: g++ over this file and lcty_exact.cpp.
#include "../../locityper_amd/csrc/lcty_exact.hpp"

#include <cstring>

extern "C" {

static void fill(lcty::exact::Model& m, uint32_t n, uint32_t tw, const uint32_t* first, const double* lp, const uint32_t* wa, const uint32_t* wb,
                 const double* ww, const uint8_t* gcb, const uint32_t* depth0, double aln0, const uint32_t* afw, uint32_t n_afw,
                 double aln_contrib, double depth_contrib) {
    m.n = n; m.tw = tw;
    m.first.assign(first, first + n + 1);
    m.locs.resize(first[n]);
    for (uint32_t t = 0; t < first[n]; t++) m.locs[t] = lcty::exact::Loc{lp[t], wa[t], wb[t]};
    m.ww.assign(ww, ww + tw); m.gcb.assign(gcb, gcb + tw); m.depth0.assign(depth0, depth0 + tw);
    m.aln0 = aln0;
    m.allele_first_w.assign(afw, afw + n_afw);
    m.aln_contrib = aln_contrib; m.depth_contrib = depth_contrib;
}

uint64_t exact_harness_depth_needed(uint32_t n, uint32_t tw, const uint32_t* first, const double* lp, const uint32_t* wa, const uint32_t* wb,
                                    const double* ww, const uint8_t* gcb, const uint32_t* depth0, const uint32_t* afw, uint32_t n_afw) {
    lcty::exact::Model m;
    fill(m, n, tw, first, lp, wa, wb, ww, gcb, depth0, 0.0, afw, n_afw, 1.0, 1.0);
    return lcty::exact::depth_needed(m);
}

// returns 1 when the search ran out of nodes (the library turns that into LCTY_ERR_SOLVER), else 0
int exact_harness_solve(uint32_t n, uint32_t tw, const uint32_t* first, const double* lp, const uint32_t* wa, const uint32_t* wb,
                        const double* ww, const uint8_t* gcb, const uint32_t* depth0, double aln0, const uint32_t* afw, uint32_t n_afw,
                        double aln_contrib, double depth_contrib, uint64_t node_limit, double rel_gap, int trace,
                        const double* lut, uint32_t ld, uint32_t gc_bins,
                        uint8_t* assign_out, double* value, double* parts, uint64_t* nodes, uint32_t* n_free) {
    lcty::exact::Model m;
    fill(m, n, tw, first, lp, wa, wb, ww, gcb, depth0, aln0, afw, n_afw, aln_contrib, depth_contrib);
    m.node_limit = node_limit; m.rel_gap = rel_gap; m.trace = trace; m.gc_bins = gc_bins;
    lcty::exact::Result r;
    lcty::exact::solve(m, lut, ld, r);
    if (assign_out) std::memcpy(assign_out, r.assign.data(), n);
    if (value) *value = r.value;
    if (parts) { parts[0] = r.aln_lik; parts[1] = r.depth_lik; }
    if (nodes) *nodes = r.nodes;
    if (n_free) *n_free = r.n_free;
    return r.out_of_nodes ? 1 : 0;
}
}
