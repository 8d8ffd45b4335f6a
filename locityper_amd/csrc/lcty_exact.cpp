// lcty_exact.cpp — branch and bound under a Lagrangian bound for the model of one (genotype, attempt) chain (lcty_exact.hpp).
//
// The reference hands an integer programme to a CPU library (highs.rs:38-100): one binary per (non-trivial read, location) with
// objective aln_contrib * ln_prob, one-hot depth variables per window with objective depth_contrib * ln_prob(depth), coupling rows; it
// asks for the optimum, fails with Error::Solver when the library does not report "optimal" (highs.rs:113-116), and decodes the
// assignment by per-read arg-max. The optimum of that model IS the assignment of largest ReadAssignment::likelihood (assgn.rs:235-237).
// Here: depth-first over the non-trivial reads (along the alleles; the location the bound's multipliers prefer first), starting from
// the best of a coordinate ascent over moves of one read and of two reads that meet in a window, pruned by a Lagrangian bound over the window counts (stated where it is set up below: lo_w is the
// depth the placed reads give window w, cap_w what the free ones could add). `node_limit` nodes without a proof -> out_of_nodes.
// Host code only: runs on the caller's pool of threads, one model per thread, like the reference's workers (solve.rs:1052-1062).
#include "lcty_exact.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <utility>

namespace lcty {
namespace exact {

namespace {
inline uint32_t mult(const Loc& l, uint32_t w) { return (l.wa == w ? 1u : 0u) + (l.wb == w ? 1u : 0u); }

// windows a read can touch, with the largest multiplicity over its locations
using Touch = std::vector<std::vector<std::pair<uint32_t, uint32_t>>>;
void touches(const Model& m, Touch& touch, std::vector<uint32_t>& cap) {
    touch.assign(m.n, {});
    cap.assign(m.tw, 0);
    for (uint32_t i = 0; i < m.n; i++) {
        for (uint32_t t = m.first[i]; t < m.first[i + 1]; t++)
            for (uint32_t w : {m.locs[t].wa, m.locs[t].wb}) {
                auto it = std::find_if(touch[i].begin(), touch[i].end(), [&](const std::pair<uint32_t, uint32_t>& x) { return x.first == w; });
                const uint32_t k = mult(m.locs[t], w);
                if (it == touch[i].end()) touch[i].push_back({w, k}); else it->second = std::max(it->second, k);
            }
        for (auto& x : touch[i]) cap[x.first] += x.second;
    }
}
// depths without the non-trivial reads (they all start at their location 0: best_start)
void depths_without(const Model& m, std::vector<int64_t>& lo, double* aln_fixed) {
    lo.assign(m.tw, 0);
    for (uint32_t w = 0; w < m.tw; w++) lo[w] = m.depth0[w];
    double a = m.aln0;
    for (uint32_t i = 0; i < m.n; i++) { const Loc& l0 = m.locs[m.first[i]]; lo[l0.wa]--; lo[l0.wb]--; a -= l0.lp; }
    *aln_fixed = a;
}
}  // namespace

uint64_t depth_needed(const Model& m) {
    Touch touch; std::vector<uint32_t> cap; std::vector<int64_t> lo; double aln_fixed;
    touches(m, touch, cap);
    depths_without(m, lo, &aln_fixed);
    uint64_t need = 0;
    for (uint32_t w = 0; w < m.tw; w++) need = std::max<uint64_t>(need, static_cast<uint64_t>(lo[w]) + cap[w] + 1);
    return need;
}

void solve(const Model& m, const double* lut, uint32_t ld, Result& out) {
    const uint32_t n = m.n, tw = m.tw;
    const std::vector<uint32_t>& first = m.first;
    const std::vector<Loc>& locs = m.locs;
    const std::vector<double>& ww = m.ww;
    Touch touch; std::vector<uint32_t> cap; std::vector<int64_t> lo; double aln_fixed;
    touches(m, touch, cap);
    depths_without(m, lo, &aln_fixed);
    auto v = [&](uint32_t w, int64_t d) -> double {                   // WindowDistr::ln_prob (distr_cache.rs:34-39)
        return ww[w] == 0.0 ? 0.0 : ww[w] * lut[static_cast<size_t>(m.gcb[w]) * ld + static_cast<size_t>(d)];
    };
    // Reads that cannot be anywhere but at their best location in an optimum: moving read i from its best location to another one
    // gains at most depth_contrib * (the largest rise any feasible depth allows the windows it leaves and the windows it enters)
    // and loses aln_contrib * (lp_best - lp_other); when the loss is larger for every other location, any assignment with the
    // read elsewhere is improved by moving it back. Such reads are fixed (they count as depth the others see), which narrows the
    // depth ranges and may fix more: repeated until nothing changes. At 1 % divergence between two alleles 19 of 20 read pairs
    // cover a difference and have a clear best location; what stays free are the pairs that match both alleles alike.
    std::vector<uint8_t> fixed(n, 0);
    auto rise = [&](uint32_t w, int dir) -> double {                   // max over feasible depths of v(d + dir) - v(d)
        if (ww[w] == 0.0) return 0.0;
        double best = -INFINITY;
        const int64_t d_lo = lo[w] + (dir < 0 ? 1 : 0), d_hi = lo[w] + cap[w] - (dir > 0 ? 1 : 0);
        for (int64_t d = d_lo; d <= d_hi; d++) best = std::max(best, v(w, d + dir) - v(w, d));
        return best == -INFINITY ? 0.0 : best;
    };
    for (bool again = true; again;) {
        again = false;
        for (uint32_t i = 0; i < n; i++) {
            if (fixed[i]) continue;
            const Loc& b0 = locs[first[i]];
            bool dominated = true;
            for (uint32_t t = first[i] + 1; t < first[i + 1] && dominated; t++) {
                const Loc& o = locs[t];
                // leaving b0's windows (their depth with the read there is >= lo + its share), entering o's; windows shared by
                // both locations cancel in the worst case as well: bounding them separately only loosens the bound
                double gain = rise(b0.wa, -1) + rise(b0.wb, -1) + rise(o.wa, 1) + rise(o.wb, 1);
                if (b0.wa == b0.wb) gain = std::max(gain, 2.0 * rise(b0.wa, -1) + rise(o.wa, 1) + rise(o.wb, 1));
                if (o.wa == o.wb) gain = std::max(gain, rise(b0.wa, -1) + rise(b0.wb, -1) + 2.0 * rise(o.wa, 1));
                if (!(m.aln_contrib * (b0.lp - o.lp) > m.depth_contrib * gain + 1e-9)) dominated = false;
            }
            if (dominated) {
                fixed[i] = 1; again = true;
                for (auto& x : touch[i]) cap[x.first] -= x.second;
                lo[b0.wa]++; lo[b0.wb]++;
                aln_fixed += b0.lp;
            }
        }
    }
    // order of the free reads: along the alleles (a window all of whose reads are placed has its exact term in the bound: a
    // wrong choice shows a few reads later, not at the end), reads of one place by the spread of their ln-probabilities
    std::vector<uint32_t> order;
    for (uint32_t i = 0; i < n; i++) if (!fixed[i]) order.push_back(i);
    auto place_of = [&](uint32_t i) {
        uint32_t best = 0xFFFFFFFFu;
        for (uint32_t t = first[i]; t < first[i + 1]; t++) for (uint32_t w : {locs[t].wa, locs[t].wb}) if (w >= 2) best = std::min(best, w);
        return best;
    };
    // windows of the alleles of a genotype lie one allele after the other: the place along the locus is the window index inside its allele
    const std::vector<uint32_t>& afw = m.allele_first_w;
    auto along = [&](uint32_t w) { uint32_t q = 0; while (q + 1 < afw.size() && w >= afw[q + 1]) q++; return w - afw[q]; };
    std::vector<uint32_t> key(n, 0);
    for (uint32_t i : order) { const uint32_t w = place_of(i); key[i] = w == 0xFFFFFFFFu ? 0u : along(w); }
    auto spread = [&](uint32_t i) { return locs[first[i]].lp - locs[first[i + 1] - 1].lp; };
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] != key[b] ? key[a] < key[b] : spread(a) > spread(b); });
    uint32_t n_free = static_cast<uint32_t>(order.size());
    // incumbent: coordinate ascent from the best start (every read at its location 0)
    std::vector<uint8_t> assign(n, 0), best_assign;
    std::vector<int64_t> dep(lo);
    for (uint32_t i : order) { dep[locs[first[i]].wa]++; dep[locs[first[i]].wb]++; }
    std::vector<int64_t> base_depth(lo);                              // `lo` moves with the search; a leaf is valued from here
    auto total = [&](const std::vector<uint8_t>& a, double* depth_lik, double* aln_lik) {
        std::vector<int64_t> d(base_depth);
        double al = aln_fixed;
        for (uint32_t i : order) { const Loc& l = locs[first[i] + a[i]]; d[l.wa]++; d[l.wb]++; al += l.lp; }
        double dl = 0.0;
        for (uint32_t w = 0; w < tw; w++) dl += v(w, d[w]);
        *depth_lik = dl; *aln_lik = al;
        return m.depth_contrib * dl + m.aln_contrib * al;
    };
    auto ascend = [&](std::vector<uint8_t>& asg, std::vector<int64_t>& dp) {
        for (bool improved = true; improved;) {
            improved = false;
            for (uint32_t i : order) {
                const Loc& cur = locs[first[i] + asg[i]];
                double best_gain = 1e-12; uint32_t best_t = asg[i];
                for (uint32_t t = 0; t < first[i + 1] - first[i]; t++) {
                    if (t == asg[i]) continue;
                    const Loc& alt = locs[first[i] + t];
                    std::pair<uint32_t, int> ch[4] = {{cur.wa, -1}, {cur.wb, -1}, {alt.wa, 1}, {alt.wb, 1}};
                    double gain = m.aln_contrib * (alt.lp - cur.lp), dd = 0.0;
                    for (int x = 0; x < 4; x++) {
                        bool seen = false; int delta = 0;
                        for (int y = 0; y < 4; y++) if (ch[y].first == ch[x].first) { if (y < x) seen = true; delta += ch[y].second; }
                        if (!seen && delta) dd += v(ch[x].first, dp[ch[x].first] + delta) - v(ch[x].first, dp[ch[x].first]);
                    }
                    gain += m.depth_contrib * dd;
                    if (gain > best_gain) { best_gain = gain; best_t = t; }
                }
                if (best_t != asg[i]) {
                    const Loc& alt = locs[first[i] + best_t];
                    dp[cur.wa]--; dp[cur.wb]--; dp[alt.wa]++; dp[alt.wb]++;
                    asg[i] = static_cast<uint8_t>(best_t); improved = true;
                }
            }
        }
    };
    // ... and moves of TWO reads that meet in a window. The optimum of these models differs from an assignment no single move improves
    // by a few hundred such pairs (one read leaves a window, another enters it: either alone pays the curvature of that window's
    // distribution, together they do not): at 10 000 read pairs the single-move ascent ends 1-2e-4 below the optimum HiGHS finds for
    // the reference's programme (tests/test_exact_highs.py), the pairs bring it to ~1e-5.
    std::vector<std::vector<uint32_t>> readers(tw);                    // free reads with a location in window w
    for (uint32_t i : order) for (auto& x : touch[i]) readers[x.first].push_back(i);
    auto move_gain = [&](uint32_t i, uint32_t from_t, uint32_t to_t, const std::vector<int64_t>& dp) -> double {
        const Loc& cur = locs[first[i] + from_t]; const Loc& alt = locs[first[i] + to_t];
        std::pair<uint32_t, int> ch[4] = {{cur.wa, -1}, {cur.wb, -1}, {alt.wa, 1}, {alt.wb, 1}};
        double dd = 0.0;
        for (int x = 0; x < 4; x++) {
            bool seen = false; int delta = 0;
            for (int y = 0; y < 4; y++) if (ch[y].first == ch[x].first) { if (y < x) seen = true; delta += ch[y].second; }
            if (!seen && delta) dd += v(ch[x].first, dp[ch[x].first] + delta) - v(ch[x].first, dp[ch[x].first]);
        }
        return m.aln_contrib * (alt.lp - cur.lp) + m.depth_contrib * dd;
    };
    auto apply_move = [&](uint32_t i, uint32_t from_t, uint32_t to_t, std::vector<int64_t>& dp) {
        const Loc& cur = locs[first[i] + from_t]; const Loc& alt = locs[first[i] + to_t];
        dp[cur.wa]--; dp[cur.wb]--; dp[alt.wa]++; dp[alt.wb]++;
    };
    auto ascend_pairs = [&](std::vector<uint8_t>& asg, std::vector<int64_t>& dp) {
        ascend(asg, dp);
        for (bool improved = true; improved;) {
            improved = false;
            for (uint32_t i : order) {
                if (fixed[i]) continue;
                const uint32_t nl = first[i + 1] - first[i];
                for (uint32_t t = 0; t < nl; t++) {
                    const uint32_t c0 = asg[i];
                    if (t == c0) continue;
                    const double g1 = move_gain(i, c0, t, dp);
                    if (g1 < -3.0) continue;                            // (a heuristic for the incumbent: what it skips costs an answer, never a wrong one)
                    apply_move(i, c0, t, dp);
                    const Loc& a = locs[first[i] + c0]; const Loc& b = locs[first[i] + t];
                    const uint32_t ws[4] = {a.wa, a.wb, b.wa, b.wb};
                    double best = 1e-9 - g1; uint32_t bj = 0xFFFFFFFFu, bt = 0;
                    for (int x = 0; x < 4; x++) {
                        bool seen = false;
                        for (int y = 0; y < x; y++) seen |= ws[y] == ws[x];
                        if (seen || ww[ws[x]] == 0.0) continue;
                        for (uint32_t j : readers[ws[x]]) {
                            if (j == i || fixed[j]) continue;
                            const uint32_t cj = asg[j], nj = first[j + 1] - first[j];
                            for (uint32_t t2 = 0; t2 < nj; t2++) {
                                if (t2 == cj) continue;
                                const double g2 = move_gain(j, cj, t2, dp);
                                if (g2 > best) { best = g2; bj = j; bt = t2; }
                            }
                        }
                    }
                    if (bj != 0xFFFFFFFFu) {
                        apply_move(bj, asg[bj], bt, dp);
                        asg[bj] = static_cast<uint8_t>(bt); asg[i] = static_cast<uint8_t>(t);
                        improved = true;
                    } else apply_move(i, t, c0, dp);                    // back
                }
            }
            if (improved) ascend(asg, dp);
        }
    };
    ascend(assign, dep);
    double dl_best, al_best;
    double incumbent = total(assign, &dl_best, &al_best);
    best_assign = assign;
    // branch and bound
    const uint64_t node_limit = m.node_limit ? m.node_limit : 20ull * 1000 * 1000;
    uint64_t nodes = 0;
    bool out_of_nodes = false;
    // The bound. With a multiplier lam_w per window the objective of any completion of the free reads is at most
    //     aln_contrib * (ln_prob placed so far) + sum over the free reads of max_t [aln_contrib * lp_t + lam over t's windows]
    //       + sum over the windows of max_{k in [0, cap_w]} [depth_contrib * v_w(lo_w + k) - lam_w * k]
    // (add and subtract lam_w x what the free reads put into window w); lam = 0 is "every free read at its best location, every window
    // at the best depth its reads could give it". The multipliers are set once, at the root, by subgradient steps that lower the
    // bound (Polyak steps towards the incumbent), and every node is bounded with them: at 10 000 read pairs the root gap falls
    // from 1.6e-2 to 6e-4 and ends ~0.45 above the optimum HiGHS proves for the reference's programme (tests/test_exact_highs.py): the
    // relaxation's own slack. The rounded multiplier solutions also feed the incumbent.
    std::vector<double> lam(tw, 0.0);
    auto wterm = [&](uint32_t w) -> double {
        if (ww[w] == 0.0) return 0.0;                                  // a window without a distribution keeps lam_w = 0
        double best = -INFINITY; const double lw = lam[w];
        for (int64_t k = 0; k <= static_cast<int64_t>(cap[w]); k++) best = std::max(best, m.depth_contrib * v(w, lo[w] + k) - lw * static_cast<double>(k));
        return best;
    };
    auto rterm = [&](uint32_t i, uint32_t* arg) -> double {
        double best = -INFINITY;
        for (uint32_t t = first[i]; t < first[i + 1]; t++) {
            const double x = m.aln_contrib * locs[t].lp + lam[locs[t].wa] + lam[locs[t].wb];
            if (x > best) { best = x; if (arg) *arg = t - first[i]; }
        }
        return best;
    };
    // ... and reads are fixed by probing: with the bound UB at the multipliers found, a solution that has read i at location t scores at
    // most UB - (the read's best term) + (the bound's terms with the read placed at t); when that is below the incumbent for every
    // location but the incumbent's own, the read stays there in every better solution. Fixed reads make the windows' ranges narrower, the bound
    // tighter, and the next round fixes more: at 10 000 read pairs all but a few hundred of the 7 800 free reads.
    // the gap the search may leave (HiGHS' mip_rel_gap): once the bound is within it of the incumbent the search below ends at its root,
    // so neither more subgradient steps nor more fixing rounds are of any use
    const double stop_gap = m.rel_gap > 0.0 && m.rel_gap < 1.0 ? m.rel_gap : 0.0;
    auto within_gap = [&](double ub) { return stop_gap > 0.0 && ub - incumbent <= 0.98 * stop_gap * std::fabs(incumbent); };
    double prev_round_ub = INFINITY;
    for (uint32_t round = 0; round < 16 && n_free > 12; round++) {
        std::vector<double> best_lam(lam), g(tw), cnt(tw), dir(tw, 0.0);
        std::vector<uint8_t> pick(best_assign);
        // how often the multipliers chose location t for read i, weighted by the step length: the running average of the subproblems'
        // solutions converges to a solution of the relaxation (Shor; Larsson, Patriksson, Stromberg 1999), which for these models is
        // integral for all but a few reads — rounded, it is a better start of the ascent than any single iterate
        std::vector<double> chosen(locs.size(), 0.0);
        double best_ub = INFINITY, theta = 1.0; uint32_t stall = 0;
        const uint32_t iters = static_cast<uint32_t>(std::min<uint64_t>(3000, 400 + n_free / 2));
        for (uint32_t it = 0; it < iters; it++) {
            std::fill(cnt.begin(), cnt.end(), 0.0);
            double ub = m.aln_contrib * aln_fixed;
            for (uint32_t i : order) {
                uint32_t t = 0;
                ub += rterm(i, &t);
                pick[i] = static_cast<uint8_t>(t);
                const Loc& l = locs[first[i] + t]; cnt[l.wa] += 1.0; cnt[l.wb] += 1.0;
            }
            const bool start_now = it % 25 == 0;
            double norm = 0.0;
            for (uint32_t w = 0; w < tw; w++) {
                g[w] = 0.0;
                if (ww[w] == 0.0) continue;
                double best = -INFINITY; int64_t kbest = 0;
                for (int64_t k = 0; k <= static_cast<int64_t>(cap[w]); k++) {
                    const double x = m.depth_contrib * v(w, lo[w] + k) - lam[w] * static_cast<double>(k);
                    if (x > best) { best = x; kbest = k; }
                }
                ub += best;
                g[w] = cnt[w] - static_cast<double>(kbest);
                norm += g[w] * g[w];
            }
            if (ub < best_ub - 1e-9) { best_ub = ub; best_lam = lam; stall = 0; }
            else if (++stall >= 20) { theta *= 0.7; stall = 0; }
            if (within_gap(best_ub)) break;
            auto start_from = [&](std::vector<uint8_t>& from) {
                std::vector<int64_t> d2(base_depth);
                for (uint32_t i : order) { const Loc& l = locs[first[i] + from[i]]; d2[l.wa]++; d2[l.wb]++; }
                ascend(from, d2);
                double dl, al;
                const double val = total(from, &dl, &al);
                if (val > incumbent) { incumbent = val; best_assign = from; dl_best = dl; al_best = al; }
            };
            if (start_now) {                                            // the multipliers' own choice of locations as a start of the ascent
                std::vector<uint8_t> from(pick);
                start_from(from);
                if (it > 0) {                                           // ... and the rounded average of their choices so far
                    std::vector<uint8_t> avg(best_assign);
                    for (uint32_t i : order) {
                        uint32_t bt = 0; double bv = -1.0;
                        for (uint32_t t = first[i]; t < first[i + 1]; t++) if (chosen[t] > bv) { bv = chosen[t]; bt = t - first[i]; }
                        avg[i] = static_cast<uint8_t>(bt);
                    }
                    start_from(avg);
                }
            }
            if (norm == 0.0 || theta < 1e-6) break;
            // deflected subgradient (Camerini, Fratta, Maffioli 1975): the direction keeps a share of the previous one whenever the two
            // point apart — the zig-zag of plain subgradient steps across the kinks of the bound is what made it creep
            double dot = 0.0, dn = 0.0;
            for (uint32_t w = 0; w < tw; w++) { dot += dir[w] * g[w]; dn += dir[w] * dir[w]; }
            const double beta = (dot < 0.0 && dn > 0.0) ? -1.5 * dot / dn : 0.0;
            double dnorm = 0.0;
            for (uint32_t w = 0; w < tw; w++) { dir[w] = g[w] + beta * dir[w]; dnorm += dir[w] * dir[w]; }
            if (dnorm == 0.0) break;
            const double step = theta * (ub - incumbent) / dnorm;
            for (uint32_t i : order) chosen[first[i] + pick[i]] += step;
            for (uint32_t w = 0; w < tw; w++) lam[w] -= step * dir[w];
        }
        lam = best_lam;
        if (!within_gap(best_ub)) {
            // the bound has gone as far as this round takes it and the incumbent is still outside the gap: the pairs (a sweep costs
            // about a hundred single-move sweeps — once per round, from the best assignment so far)
            std::vector<uint8_t> from(best_assign);
            std::vector<int64_t> d2(base_depth);
            for (uint32_t i : order) { const Loc& l = locs[first[i] + from[i]]; d2[l.wa]++; d2[l.wb]++; }
            ascend_pairs(from, d2);
            double dl, al;
            const double val = total(from, &dl, &al);
            if (val > incumbent) { incumbent = val; best_assign = from; dl_best = dl; al_best = al; }
        }
        // reduced-cost fixing at these multipliers
        double ub = m.aln_contrib * aln_fixed;
        std::vector<double> rbest(n, 0.0);
        for (uint32_t i : order) { rbest[i] = rterm(i, nullptr); ub += rbest[i]; }
        for (uint32_t w = 0; w < tw; w++) ub += wterm(w);
        const double floor_val = incumbent - (1e-9 * std::fabs(incumbent) + 1e-9);
        std::vector<uint32_t> still;
        uint32_t newly = 0;
        for (uint32_t i : order) {
            const uint32_t b = best_assign[i];
            bool only = true;
            for (uint32_t t = first[i]; t < first[i + 1] && only; t++) {
                if (t - first[i] == b) continue;
                // the bound with the read AT t (what the search computes one level down): its ln-probability, and the windows it can
                // touch with the read counted where t puts it and no longer among what the free reads could add — never above
                // the reduced-cost form "the read's term at t" (a window's term falls by at least lam x what the read adds)
                double x = m.aln_contrib * locs[t].lp;
                for (auto& tw_ : touch[i]) {
                    const uint32_t w = tw_.first;
                    if (ww[w] == 0.0) continue;
                    const int64_t add = static_cast<int64_t>(mult(locs[t], w)), room = static_cast<int64_t>(cap[w]) - tw_.second;
                    double best = -INFINITY; const double lw = lam[w];
                    for (int64_t k = 0; k <= room; k++) best = std::max(best, m.depth_contrib * v(w, lo[w] + add + k) - lw * static_cast<double>(k));
                    x += best - wterm(w);
                }
                if (!(ub - rbest[i] + x < floor_val)) only = false;
            }
            if (!only) { still.push_back(i); continue; }
            const Loc& l = locs[first[i] + b];
            fixed[i] = 2; newly++;
            for (auto& x : touch[i]) cap[x.first] -= x.second;
            lo[l.wa]++; lo[l.wb]++; base_depth[l.wa]++; base_depth[l.wb]++;
            aln_fixed += l.lp;
        }
        order.swap(still);
        n_free = static_cast<uint32_t>(order.size());
        if (m.trace)
            fprintf(stderr, "[lcty exact] chain %u round %u: bound %.6f, incumbent %.6f, %u reads fixed by reduced costs, %u free\n", m.chain, round, ub, incumbent, newly, n_free);
        if (within_gap(ub)) break;
        // another round starts from these multipliers with full steps again: worth it while reads get fixed or the bound still moves
        if (newly == 0 && !(ub < prev_round_ub - 2e-6 * std::fabs(incumbent))) break;
        prev_round_ub = ub;
    }
    if (n_free > 0) {
        // whatever the rounds left: the answer itself from the pairs once more — inside the gap the search below ends at its root, and the
        // reference's solver returns the optimum where its relaxation is integral (it is for these models: HiGHS ends at the root node)
        std::vector<uint8_t> from(best_assign);
        std::vector<int64_t> d2(base_depth);
        for (uint32_t i : order) { const Loc& l = locs[first[i] + from[i]]; d2[l.wa]++; d2[l.wb]++; }
        ascend_pairs(from, d2);
        double dl, al;
        const double val = total(from, &dl, &al);
        if (val > incumbent) { incumbent = val; best_assign = from; dl_best = dl; al_best = al; }
    }
    std::vector<double> rmax(n, 0.0);
    std::vector<uint8_t> first_try(n, 0);                              // the location the multipliers prefer is explored first
    double free_best = 0.0;                                            // sum over the free reads of their terms of the bound
    for (uint32_t i : order) { uint32_t t = 0; rmax[i] = rterm(i, &t); first_try[i] = static_cast<uint8_t>(t); free_best += rmax[i]; }
    auto explored = [&](uint32_t i, uint32_t e) -> uint32_t { return e == 0 ? first_try[i] : (e - 1 < first_try[i] ? e - 1 : e); };
    double aln_sum = aln_fixed, win_sum = 0.0;
    std::vector<double> wmax(tw);
    for (uint32_t w = 0; w < tw; w++) { wmax[w] = wterm(w); win_sum += wmax[w]; }
    // HiGHS stops a search — and reports "optimal" — when the bound of what is left is within mip_rel_gap (1e-4 by default) of the
    // incumbent; the reference leaves that option alone (highs.rs:103-110). Subtrees that cannot beat the incumbent by more than the
    // gap are left out; rel_gap = 0 is a proof of optimality.
    const double rel_gap = m.rel_gap > 0.0 && m.rel_gap < 1.0 ? m.rel_gap : 0.0;
    const double root_bound = m.aln_contrib * aln_sum + free_best + win_sum;
    if (!m.dump_path.empty()) {
        // the model as the search sees it (scripts/exact_probe.py --dump): text, one item per line
        FILE* f = fopen(m.dump_path.c_str(), "w");
        if (f) {
            fprintf(f, "%u %u %u %.17g %.17g %.17g\n", n, tw, ld, m.aln_contrib, m.depth_contrib, aln_fixed);
            for (uint32_t w = 0; w < tw; w++) fprintf(f, "W %lld %u %.17g %u\n", static_cast<long long>(base_depth[w]), cap[w], ww[w], static_cast<unsigned>(m.gcb[w]));
            for (uint32_t i = 0; i < n; i++) {
                fprintf(f, "R %u %u", static_cast<unsigned>(fixed[i]), first[i + 1] - first[i]);
                for (uint32_t t = first[i]; t < first[i + 1]; t++) fprintf(f, " %.17g %u %u", locs[t].lp, locs[t].wa, locs[t].wb);
                fprintf(f, "\n");
            }
            for (uint32_t g = 0; g < m.gc_bins; g++) {
                bool used = false;
                for (uint32_t w = 0; w < tw; w++) used |= ww[w] != 0.0 && m.gcb[w] == g;
                if (!used) continue;
                fprintf(f, "L %u", g);
                for (uint32_t d = 0; d < ld; d++) fprintf(f, " %.17g", lut[static_cast<size_t>(g) * ld + d]);
                fprintf(f, "\n");
            }
            fprintf(f, "I %.17g\n", incumbent);
            fclose(f);
        }
    }
    if (m.trace)
        fprintf(stderr, "[lcty exact] chain %u: %u non-trivial reads, %u free; incumbent %.6f, root bound %.6f (gap %.3e relative)\n", m.chain, n, n_free,
                incumbent, root_bound, (root_bound - incumbent) / std::fabs(incumbent));
    // depth-first, iterative (a locus can have many thousands of non-trivial reads: no recursion)
    std::vector<uint8_t> cur_assign(best_assign), entered(n, 0), applied(n, 0);      // the reads fixed above keep their locations
    std::vector<uint32_t> next_t(n, 0);
    std::vector<double> keep_ws(n), keep_al(n);
    std::vector<std::vector<std::pair<uint32_t, double>>> saved(n);
    int64_t level = 0;
    while (level >= 0 && n_free) {
        if (static_cast<uint32_t>(level) == n_free) {                  // a leaf: the value as ReadAssignment::likelihood sums it
            if (++nodes > node_limit) out_of_nodes = true;
            double dl, al;
            const double val = total(cur_assign, &dl, &al);
            if (val > incumbent) { incumbent = val; best_assign = cur_assign; dl_best = dl; al_best = al; }
            level--;
            continue;
        }
        const uint32_t i = order[level], nloc = first[i + 1] - first[i];
        if (!entered[level]) {
            // the read leaves the free set: what it could have added to its windows goes, its best ln-probability too
            if (++nodes > node_limit) out_of_nodes = true;
            entered[level] = 1; applied[level] = 0; next_t[level] = 0;
            for (auto& x : touch[i]) cap[x.first] -= x.second;
            free_best -= rmax[i];
        }
        if (applied[level]) {                                           // back from (or past) the location tried last
            const Loc& l = locs[first[i] + explored(i, next_t[level] - 1)];
            for (auto& sv : saved[level]) wmax[sv.first] = sv.second;
            lo[l.wa]--; lo[l.wb]--;
            win_sum = keep_ws[level]; aln_sum = keep_al[level];
            applied[level] = 0;
        }
        if (next_t[level] == nloc || out_of_nodes) {
            free_best += rmax[i];
            for (auto& x : touch[i]) cap[x.first] += x.second;
            entered[level] = 0;
            level--;
            continue;
        }
        const uint32_t t = explored(i, next_t[level]++);
        const Loc& l = locs[first[i] + t];
        lo[l.wa]++; lo[l.wb]++;
        saved[level].clear();
        double ws_new = win_sum;
        for (auto& x : touch[i]) {
            const double best = wterm(x.first);
            saved[level].push_back({x.first, wmax[x.first]});
            ws_new += best - wmax[x.first]; wmax[x.first] = best;
        }
        applied[level] = 1; keep_ws[level] = win_sum; keep_al[level] = aln_sum;
        const double bound = m.aln_contrib * (aln_sum + l.lp) + free_best + ws_new;
        // a subtree is left out when it cannot beat the incumbent by more than the rounding of two long sums
        if (bound > incumbent + std::max(rel_gap, 1e-12) * std::fabs(incumbent) + 1e-10) {
            win_sum = ws_new; aln_sum += l.lp; cur_assign[i] = static_cast<uint8_t>(t);
            level++;
        }
    }
    out.assign = best_assign;
    out.value = incumbent; out.depth_lik = dl_best; out.aln_lik = al_best;
    out.nodes = nodes; out.out_of_nodes = out_of_nodes; out.n_free = n_free;
}

}  // namespace exact
}  // namespace lcty
