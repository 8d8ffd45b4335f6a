// lcty_score.hip — AllAlignments::load (src/model/locs.rs:1085-1185, 1237-1288, no alignment
// recovery) as ONE fused gfx950 kernel: one 64-lane wavefront per read pair.
//
//   pass 1  lanes over BAM records: CIGAR -> op counts -> ErrorProfile::ln_prob, edit distance
//           (aln.rs:301-317, err_prof.rs:73-79, 212-221); results parked in LDS (24 B / record)
//   thresholds  lanes 0/1: EditDistCache + poor-complexity relaxation (locs.rs:529-536)
//   pass 2  per-end best edit / best ln-prob (wave reductions), saved flag (locs.rs:310-314)
//   K2      unique k-mers of both mates from the 2-bit stream: wave-parallel hash probes,
//           ballot, greedy non-overlap walk over the hit mask (locs.rs:968-1002)
//   sort    LDS counting sort of the saved records by (contig, read end)
//   pass 3a lane per contig: (ln_prob desc) order, 128-bp dedupe (locs.rs:321-342), in_bounds
//           (1008-1014), pair enumeration -> best + kept count (746-799) -> matrix row (1203-1212)
//   pass 3b selection-emit of the kept PairAlignments, contig-ascending, into the arena
//
// Launch: 64-thread workgroups (one wave), grid-strided over pairs; dynamic LDS sized from the
// largest record count of any pair in the batch. All LDS traffic is wave-private.
#include "lcty_objects.hpp"

namespace lcty {

struct RecLds {
    double ln_prob;
    uint32_t start, end, edit;
    uint16_t contig, flags;
};
static_assert(sizeof(RecLds) == 24, "RecLds layout");

enum : uint16_t {
    RF_REVERSE = 1, RF_SAVED = 4, RF_SKIP = 8, RF_UNMAPPED = 16, RF_BAD = 32, RF_PRIMARY = 64,
};

constexpr uint32_t NONE32 = 0xFFFFFFFFu;

__device__ inline uint32_t wave_min_u32(uint32_t v) {
    for (int o = WAVE / 2; o > 0; o >>= 1) v = min(v, static_cast<uint32_t>(__shfl_xor(static_cast<int>(v), o)));
    return v;
}
__device__ inline double wave_max_f64(double v) {
    for (int o = WAVE / 2; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ inline uint32_t wave_sum_u32(uint32_t v) {
    for (int o = WAVE / 2; o > 0; o >>= 1) v += static_cast<uint32_t>(__shfl_xor(static_cast<int>(v), o));
    return v;
}
// exclusive prefix sum across the wave; *total = wave sum
__device__ inline uint32_t wave_excl_scan_u32(uint32_t v, int lane, uint32_t* total) {
    uint32_t x = v;
    for (int o = 1; o < WAVE; o <<= 1) {
        const uint32_t y = static_cast<uint32_t>(__shfl_up(static_cast<int>(x), o));
        if (lane >= o) x += y;
    }
    *total = static_cast<uint32_t>(__shfl(static_cast<int>(x), WAVE - 1));
    return x - v;
}

// InsertDistr::ln_prob -> LinearCache::ln_pmf (insertsz.rs:153-155, lincache.rs:41-48)
// cold path: insert sizes beyond the cached range are evaluated directly (nbinom.rs:128-131)
__device__ __noinline__ double nbinom_ln_pmf_direct(double n, double lnq, double lnpmf_const, uint32_t sz) {
    const double x = static_cast<double>(sz);
    return lnpmf_const + lgamma(n + x) - lgamma(x + 1.0) + x * lnq;
}
__device__ inline double insert_lnprob(const LocusView& L, uint32_t sz) {
    if (sz < L.ins_lut_size) return L.ins_lut[sz];
    return nbinom_ln_pmf_direct(L.ins_n, L.ins_lnq, L.ins_lnpmf_const, sz);
}

// UniqueKmers::calculate_read_weight for one mate (locs.rs:976-993): count of non-overlapping hits.
__device__ inline uint32_t mate_unique_kmers(const LocusView& L, const ReadsView& R, uint64_t mate, int lane) {
    const uint32_t len = R.mate_len[mate];
    const uint32_t k = L.k;
    if (len < k) return 0;
    const uint64_t off = R.mate_off[mate];
    const uint64_t* w64 = reinterpret_cast<const uint64_t*>(R.bases2) + (off >> 5);
    const uint32_t* nm = R.nmask + (off >> 5);
    const uint32_t nk = len + 1 - k;
    uint32_t count = 0, next_allowed = 0;
    for (uint32_t base = 0; base < nk; base += WAVE) {
        const uint32_t q = base + lane;
        bool hit = false;
        if (q < nk) {
            if (window_has_n(nm, q, k)) hit = L.undef_in_set != 0;       // UNDEF k-mer (kmers.rs:184-190)
            else hit = kset_contains(L.kset, L.kset_mask, canonical_kmer_2bit(w64, q, k));
        }
        unsigned long long m = __ballot(hit);
        // greedy walk: take a hit, then skip the next k-1 k-mers (`kmers_iter.nth(k_2)`, locs.rs:988)
        if (next_allowed > base) {
            const uint32_t sh = next_allowed - base;
            m = sh >= 64 ? 0ull : (m >> sh) << sh;
        }
        while (m) {
            const uint32_t b = static_cast<uint32_t>(__ffsll(static_cast<long long>(m))) - 1u;
            count = count == 0xFFFFu ? count : count + 1;                 // saturating_add
            next_allowed = base + b + k;
            const uint32_t sh = b + k;
            m = sh >= 64 ? 0ull : (m >> sh) << sh;
        }
    }
    return count;
}

struct AlnRef {
    double lp;
    uint32_t start, end, idx;
    bool rev;
};

struct PairCtx {
    const LocusView* L;
    const RecLds* rec;
    const uint16_t* ord1;
    const uint16_t* ord2;
    uint32_t k1, k2;
    double best0, best1;        // best_lik per end (normalize_probs, locs.rs:358-360)
    double unm_ins_penalty;
    bool paired;

    __device__ inline AlnRef get1(uint32_t i) const {
        const uint32_t v = ord1[i];
        const RecLds& r = rec[v];
        return AlnRef{r.ln_prob - best0, r.start, r.end, v, (r.flags & RF_REVERSE) != 0};
    }
    __device__ inline AlnRef get2(uint32_t j) const {
        const uint32_t v = ord2[j];
        const RecLds& r = rec[v];
        return AlnRef{r.ln_prob - best1, r.start, r.end, v, (r.flags & RF_REVERSE) != 0};
    }
    __device__ inline double pair_prob(const AlnRef& a1, const AlnRef& a2) const {
        // paired_prob (aln.rs:236-238) with furthest_distance (interv.rs:179-185)
        const uint32_t insert = max(a1.end, a2.end) - min(a1.start, a2.start);
        return a1.lp + a2.lp + insert_lnprob(*L, insert);
    }

    // Calls f(prob, order, aln1 or idx NONE32, aln2 or idx NONE32) for every PairAlignment pushed by
    // identify_contig_pair_alns (locs.rs:762-791) in push order, or — single-end — by
    // identify_single_end_alignments (locs.rs:890-901).
    template <typename F>
    __device__ inline void enumerate(F&& f) const {
        const AlnRef none{0.0, 0, 0, NONE32, false};
        if (!paired) {
            for (uint32_t i = 0; i < k1; i++) { const AlnRef a = get1(i); f(a.lp, i, a, none); }
            return;
        }
        for (uint32_t i = 0; i < k1; i++) {
            const AlnRef a1 = get1(i);
            double m1 = -INFINITY;
            for (uint32_t j = 0; j < k2; j++) {
                const AlnRef a2 = get2(j);
                if (a1.rev != a2.rev) {
                    const double prob = pair_prob(a1, a2);
                    if (isfinite(prob)) { m1 = fmax(m1, prob); f(prob, i * (k2 + 1) + j, a1, a2); }
                }
            }
            const double alone = a1.lp + unm_ins_penalty;
            if (alone >= m1) f(alone, i * (k2 + 1) + k2, a1, none);
        }
        for (uint32_t j = 0; j < k2; j++) {
            const AlnRef a2 = get2(j);
            double m2 = -INFINITY;
            for (uint32_t i = 0; i < k1; i++) {
                const AlnRef a1 = get1(i);
                if (a1.rev != a2.rev) {
                    const double prob = pair_prob(a1, a2);
                    if (isfinite(prob)) m2 = fmax(m2, prob);
                }
            }
            const double alone = a2.lp + unm_ins_penalty;
            if (alone >= m2) f(alone, k1 * (k2 + 1) + j, none, a2);
        }
    }
};

// in-place (ln_prob desc, record index asc) insertion sort + 128-bp-bin dedupe of one (contig, end) group.
// Returns the number of kept alignments, *inb |= any kept alignment inside the central region.
__device__ inline uint32_t sort_dedupe(const RecLds* rec, uint16_t* ord, uint32_t n, uint32_t boundary,
                                       uint32_t contig_len, bool* inb) {
    for (uint32_t i = 1; i < n; i++) {
        const uint16_t v = ord[i];
        const double lpv = rec[v].ln_prob;
        uint32_t j = i;
        while (j > 0) {
            const uint16_t u = ord[j - 1];
            const double lpu = rec[u].ln_prob;
            if (lpu < lpv || (lpu == lpv && u > v)) { ord[j] = u; j--; } else break;
        }
        ord[j] = v;
    }
    uint32_t kept = 0;
    for (uint32_t t = 0; t < n; t++) {
        const uint16_t v = ord[t];
        const uint32_t bin = rec[v].start >> 7;                 // STEP_PWR, locs.rs:174,183
        bool dup = false;
        for (uint32_t u = 0; u < kept; u++) dup |= (rec[ord[u]].start >> 7) == bin;
        if (!dup) {
            ord[kept++] = v;
            const uint32_t mid = (rec[v].start + rec[v].end) / 2;
            if (boundary <= mid && mid < contig_len - boundary) *inb = true;    // in_bounds, locs.rs:1008-1014
        }
    }
    return kept;
}

__global__ __launch_bounds__(WAVE) void score_reads_kernel(const LocusView L, const ReadsView R, const uint32_t max_recs, const uint32_t dbg) {
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t A = L.n_alleles;
    RecLds* rec = reinterpret_cast<RecLds*>(smem);
    uint32_t* hist = reinterpret_cast<uint32_t*>(rec + max_recs);          // [2A] counts -> offsets -> group ends
    uint16_t* order = reinterpret_cast<uint16_t*>(hist + 2 * A);           // [max_recs]
    uint8_t* kk1 = reinterpret_cast<uint8_t*>(order + ((max_recs + 1) & ~1u));   // [A] kept first-end alns (<= 10)
    uint8_t* kk2 = kk1 + A;                                                // [A]
    uint8_t* cnt8 = kk2 + A;                                               // [A] emitted PairAlignments (<= 10)
    const int lane = threadIdx.x;

    for (uint64_t p = blockIdx.x; p < R.n_pairs; p += gridDim.x) {
        const uint64_t a0 = R.aln_off[p];
        const uint32_t n = static_cast<uint32_t>(R.aln_off[p + 1] - a0);
        const uint32_t* cig = R.cigar + R.cigar_off[p];
        double* mrow = R.matrix + p * A;

        // ---------------- pass 1: records -> LDS ----------------
        uint32_t j2 = NONE32, j3 = NONE32;
        for (uint32_t base = 0; base < n; base += WAVE) {
            const uint32_t idx = base + lane;
            bool primary = false;
            if (idx < n) {
                const uint4 raw = reinterpret_cast<const uint4*>(R.recs)[a0 + idx];
                const uint32_t pos = raw.x, contig = raw.y & 0xFFFFu, bflags = raw.y >> 16, nc = raw.z, rel = raw.w;
                primary = (bflags & (LCTY_FLAG_SECONDARY | LCTY_FLAG_SUPPL)) == 0;
                uint16_t fl = (bflags & LCTY_FLAG_REVERSE) ? RF_REVERSE : 0;
                if (primary) fl |= RF_PRIMARY;
                if (bflags & LCTY_FLAG_UNMAPPED) fl |= RF_UNMAPPED;
                uint32_t matches = 0, mism = 0, ins = 0, del = 0, left = 0, right = 0;
                if (nc == 0) fl |= RF_SKIP;
                if (contig >= A) fl |= RF_BAD;
                for (uint32_t i = 0; i < nc; i++) {
                    const uint32_t w = cig[rel + i];
                    uint32_t op = w & 15u;
                    const uint32_t len = w >> 4;
                    if (op == LCTY_CIGAR_H && (i == 0 || i + 1 == nc)) {
                        if (primary) fl |= RF_BAD;              // assert!(!cigar.has_hard_clipping()), locs.rs:526
                        op = LCTY_CIGAR_S;                       // hard_to_soft, cigar.rs:309-320
                    }
                    switch (op) {
                        case LCTY_CIGAR_EQ: matches += len; break;
                        case LCTY_CIGAR_X: mism += len; break;
                        case LCTY_CIGAR_D: del += len; break;
                        case LCTY_CIGAR_I: ins += len; break;
                        case LCTY_CIGAR_S:
                            if (i == 0) left = len;              // soft_clipping, cigar.rs:519-527
                            if (i + 1 == nc) right = len;
                            break;
                        default: fl |= RF_BAD;                   // panic!("Unsupported CIGAR operation"), aln.rs:311
                    }
                }
                const uint32_t ref_len = matches + mism + del;
                const uint32_t clen = (fl & RF_BAD) ? 0u : L.allele_len[contig];
                const uint32_t end = pos + ref_len;
                const uint32_t clip = min(left, pos) + min(right, clen > end ? clen - end : 0u);   // limited_clipping, aln.rs:288-296
                const uint32_t common = mism + ins + clip;                                          // err_prof.rs:73-79
                RecLds r;
                r.ln_prob = L.lp[0] * static_cast<double>(matches) + L.lp[1] * static_cast<double>(mism)
                          + L.lp[2] * static_cast<double>(ins) + L.lp[3] * static_cast<double>(del)
                          + L.lp[4] * static_cast<double>(clip);                                    // err_prof.rs:212-221
                r.start = pos; r.end = end; r.edit = common + del;
                r.contig = static_cast<uint16_t>(contig); r.flags = fl;
                rec[idx] = r;
            }
            unsigned long long b = __ballot(primary && idx > 0);
            if (b && j2 == NONE32) {
                const uint32_t f = static_cast<uint32_t>(__ffsll(static_cast<long long>(b))) - 1u;
                j2 = base + f;
                b &= ~((2ull << f) - 1ull);
            }
            if (b && j2 != NONE32 && j3 == NONE32) j3 = base + static_cast<uint32_t>(__ffsll(static_cast<long long>(b))) - 1u;
        }
        __syncthreads();
        if (dbg == 1) { if (lane == 0) R.status[p] = static_cast<uint8_t>(j2 + rec[0].edit); continue; }   // ablation (LCTY_DBG)
        // records of this pair: end 0 = [0, j2), end 1 = [j2, n_eff) (locs.rs:1119-1131)
        const uint32_t n_eff = L.is_paired ? min(n, j3) : min(n, j2);
        const uint32_t split = min(j2, n_eff);

        // ---------------- thresholds (lanes 0 / 1) ----------------
        uint32_t my_good = 0, my_thr = NONE32, my_pass = NONE32, my_state = 0;   // state: 1 examined-ok, 2 primary saved, 4 error
        if (lane < 2) {
            const uint32_t e = lane;
            const uint32_t pidx = e ? j2 : 0u;
            const bool exists = e == 0 ? n_eff > 0 : (L.is_paired && j2 < n_eff);
            if (e == 1 && !L.is_paired) {
                my_state = 3;                                   // single-end: second end is vacuously fine
            } else if (!exists) {
                my_state = 4;                                   // expect("Cannot read any more records"), locs.rs:509
            } else {
                const RecLds pr = rec[pidx];
                const uint32_t read_len = R.mate_len[2 * p + e];
                if (read_len == 0 || !(pr.flags & RF_PRIMARY)) my_state = 4;        // locs.rs:511-517 / LaggedReader assert
                else if (pr.flags & RF_UNMAPPED) my_state = 0;                      // locs.rs:520-523
                else if (pr.flags & (RF_BAD | RF_SKIP)) my_state = 4;
                else {
                    const uint2 gp = L.edit_lut[min(read_len, L.edit_lut_size - 1)];
                    uint32_t good = gp.x, passable = gp.y, thr = good;
                    double compl_v = 1.0;
                    if (L.short_reads) {                         // neighb_complexity, windows.rs:447-452, 696-698
                        const uint32_t mid = (pr.start + pr.end) / 2;
                        const uint32_t npos = L.ci_off[pr.contig + 1] - L.ci_off[pr.contig];
                        const uint32_t i = min(mid > L.half_neighb ? mid - L.half_neighb : 0u, npos - 1);
                        compl_v = static_cast<double>(L.compl_cnt[L.ci_off[pr.contig] + i]) * L.compl_mult;
                    }
                    if (compl_v <= L.poor_compl) {               // locs.rs:533-536
                        thr = max(good, static_cast<uint32_t>(L.poor_compl_edit * static_cast<double>(read_len)));
                        passable += thr - good;
                    }
                    my_good = good; my_thr = thr; my_pass = passable;
                    my_state = 1u | (pr.edit <= passable ? 2u : 0u);
                }
            }
        }
        const uint32_t good0 = __shfl(my_good, 0), thr0 = __shfl(my_thr, 0), pass0 = __shfl(my_pass, 0), st0 = __shfl(my_state, 0);
        const uint32_t good1 = __shfl(my_good, 1), thr1 = __shfl(my_thr, 1), pass1 = __shfl(my_pass, 1), st1 = __shfl(my_state, 1);

        // ---------------- pass 2: per-end best edit / ln-prob, saved flags ----------------
        uint32_t be0 = NONE32, be1 = NONE32, bad0 = 0, bad1 = 0;
        double bl0 = -INFINITY, bl1 = -INFINITY;
        for (uint32_t idx = lane; idx < n_eff; idx += WAVE) {
            RecLds& r = rec[idx];
            const uint32_t e = idx >= split ? 1u : 0u;
            // a record is examined only if its end's primary was pushed (locs.rs:539-558)
            const bool examined = e == 0 ? (st0 & 3u) == 3u : ((st0 & 3u) == 3u && (st1 & 3u) == 3u && L.is_paired);
            if (!examined || (r.flags & RF_SKIP)) continue;
            if (r.flags & RF_BAD) { if (e) bad1 = 1; else bad0 = 1; continue; }
            if (e == 0) { be0 = min(be0, r.edit); bl0 = fmax(bl0, r.ln_prob); }
            else { be1 = min(be1, r.edit); bl1 = fmax(bl1, r.ln_prob); }
            if (r.edit <= (e ? pass1 : pass0)) r.flags |= RF_SAVED;
        }
        be0 = wave_min_u32(be0); be1 = wave_min_u32(be1);
        bl0 = wave_max_f64(bl0); bl1 = wave_max_f64(bl1);
        bad0 = wave_sum_u32(bad0); bad1 = wave_sum_u32(bad1);
        // end 1 is only looked at when end 0 is well mapped (locs.rs:1125-1132)
        const bool wm0 = (st0 & 3u) == 3u && be0 <= (L.strict_subset ? pass0 : thr0) && !bad0;
        const bool err = (st0 & 4u) || bad0 || (wm0 && ((st1 & 4u) || bad1));
        if (err) {
            if (lane == 0) atomicMax(R.err_flag, static_cast<uint32_t>(LCTY_ERR_INVALID_DATA));
        }
        const bool wm1 = !L.is_paired || ((st1 & 3u) == 3u && be1 <= (L.strict_subset ? pass1 : thr1));
        bool accepted = wm0 && wm1 && !err;
        double weight = 1.0;
        if (accepted) {
            weight *= be0 <= good0 ? 1.0 : sqrt(static_cast<double>(good0) / static_cast<double>(be0));      // locs.rs:565
            if (L.is_paired) weight *= be1 <= good1 ? 1.0 : sqrt(static_cast<double>(good1) / static_cast<double>(be1));
        }
        uint8_t status = LCTY_READ_POORLY_MAPPED;
        uint32_t total_cnt = 0;
        double unmapped_prob = 0.0;
        uint64_t pa_base = 0;

        if (accepted) {
            // ---------------- counting sort of saved records by (contig, end) ----------------
            for (uint32_t i = lane; i < 2 * A; i += WAVE) hist[i] = 0;
            __syncthreads();
            for (uint32_t idx = lane; idx < n_eff; idx += WAVE) {
                const RecLds& r = rec[idx];
                if (r.flags & RF_SAVED) atomicAdd(&hist[2u * r.contig + (idx >= split ? 1u : 0u)], 1u);
            }
            __syncthreads();
            {
                const uint32_t per = (2 * A + WAVE - 1) / WAVE;
                const uint32_t lo = min(lane * per, 2 * A), hi = min(lo + per, 2 * A);
                uint32_t s = 0;
                for (uint32_t i = lo; i < hi; i++) s += hist[i];
                uint32_t tot;
                uint32_t run = wave_excl_scan_u32(s, lane, &tot);
                for (uint32_t i = lo; i < hi; i++) { const uint32_t c = hist[i]; hist[i] = run; run += c; }
            }
            __syncthreads();
            for (uint32_t idx = lane; idx < n_eff; idx += WAVE) {
                const RecLds& r = rec[idx];
                if (r.flags & RF_SAVED) {
                    const uint32_t pos = atomicAdd(&hist[2u * r.contig + (idx >= split ? 1u : 0u)], 1u);
                    order[pos] = static_cast<uint16_t>(idx);
                }
            }
            __syncthreads();
            if (dbg == 2) { if (lane == 0) R.status[p] = static_cast<uint8_t>(order[0]); continue; }          // ablation

            // ---------------- K2: unique k-mers -> read weight (locs.rs:968-1002) ----------------
            // (evaluated after the in-bounds test in the reference; it has no side effects, so the
            //  order is irrelevant for the result)
            const uint32_t uk0 = dbg == 3 ? 5u : mate_unique_kmers(L, R, 2 * p, lane);
            const uint32_t uk1 = dbg == 3 ? 5u : (L.is_paired && R.mate_len[2 * p + 1] ? mate_unique_kmers(L, R, 2 * p + 1, lane) : 0u);
            const uint32_t paired_count = (uk0 + uk1) & 0xFFFFu;
            double kw = L.weight_interc + static_cast<double>(paired_count) * L.weight_mult;
            kw = kw < 0.0 ? 0.0 : (kw > 1.0 ? 1.0 : kw);
            weight *= kw;
            const uint32_t max_alns = weight >= L.min_weight ? LCTY_MAX_USED_ALNS : LCTY_MAX_UNUSED_ALNS;   // locs.rs:1268
            const double unm_ins_penalty = L.unmapped_penalty + L.insert_penalty;                         // locs.rs:816
            unmapped_prob = L.is_paired ? weight * (2.0 * L.unmapped_penalty + L.insert_penalty)           // locs.rs:866
                                        : weight * L.unmapped_penalty;                                     // locs.rs:908

            // ---------------- pass 3a: per contig sort/dedupe, best + count ----------------
            bool inb = false;
            for (uint32_t c0 = 0; c0 < A; c0 += WAVE) {
                const uint32_t c = c0 + lane;
                if (c < A) {
                    const uint32_t s1 = c ? hist[2 * c - 1] : 0u, e1 = hist[2 * c], e2 = hist[2 * c + 1];
                    const uint32_t clen = L.allele_len[c];
                    const uint32_t K1 = sort_dedupe(rec, order + s1, e1 - s1, L.boundary, clen, &inb);
                    const uint32_t K2 = sort_dedupe(rec, order + e1, e2 - e1, L.boundary, clen, &inb);
                    const uint32_t k1 = min(K1, max_alns), k2 = min(K2, max_alns);       // locs.rs:842-851
                    kk1[c] = static_cast<uint8_t>(k1); kk2[c] = static_cast<uint8_t>(k2);
                    double mval = unmapped_prob;
                    uint32_t cnt = 0;
                    if (k1 + k2 > 0) {
                        PairCtx pc{&L, rec, order + s1, order + e1, k1, k2, bl0, bl1, unm_ins_penalty, L.is_paired != 0};
                        double best = -INFINITY;
                        pc.enumerate([&](double prob, uint32_t, const AlnRef&, const AlnRef&) { best = fmax(best, prob); });
                        const double thresh = best - L.prob_diff;                        // locs.rs:796
                        uint32_t ge = 0;
                        pc.enumerate([&](double prob, uint32_t, const AlnRef&, const AlnRef&) { ge += prob >= thresh; });
                        cnt = min(ge, max_alns);                                         // locs.rs:797
                        if (cnt) mval = best * weight;                                   // locs.rs:861-863, 621-629
                    }
                    cnt8[c] = static_cast<uint8_t>(cnt);
                    total_cnt += cnt;
                    mrow[c] = mval;
                }
            }
            const bool any_inb = __ballot(inb) != 0ull;
            total_cnt = wave_sum_u32(total_cnt);
            const bool edit_good = be0 <= thr0 && (!L.is_paired || be1 <= thr1);          // best_edit_is_good, locs.rs:293-295
            if (!any_inb) { status = LCTY_READ_OUT_OF_BOUNDS; accepted = false; }
            else if (!edit_good) { status = LCTY_READ_POORLY_MAPPED; accepted = false; }
            else status = weight >= L.min_weight ? LCTY_READ_GOOD : LCTY_READ_FEW_KMERS;     // locs.rs:1277-1285

            if (accepted) {
                if (lane == 0) {
                    pa_base = atomicAdd(R.pa_count, static_cast<unsigned long long>(total_cnt));
                    if (pa_base + total_cnt > R.pa_cap) atomicMax(R.err_flag, static_cast<uint32_t>(LCTY_ERR_RUNTIME));
                    R.uniq_kmers[2 * p] = static_cast<uint16_t>(uk0);
                    R.uniq_kmers[2 * p + 1] = static_cast<uint16_t>(uk1);
                }
                pa_base = __shfl(pa_base, 0);
                const bool room = pa_base + total_cnt <= R.pa_cap;
                // ---------------- pass 3b: emit PairAlignments, contig-ascending ----------------
                uint32_t run = 0;
                for (uint32_t c0 = 0; c0 < A; c0 += WAVE) {
                    const uint32_t c = c0 + lane;
                    const uint32_t cnt = c < A ? cnt8[c] : 0u;
                    uint32_t tot;
                    const uint32_t my_off = run + wave_excl_scan_u32(cnt, lane, &tot);
                    run += tot;
                    if (cnt && room && dbg != 4) {
                        const uint32_t s1 = c ? hist[2 * c - 1] : 0u, e1 = hist[2 * c];
                        PairCtx pc{&L, rec, order + s1, order + e1, kk1[c], kk2[c], bl0, bl1, unm_ins_penalty, L.is_paired != 0};
                        PairAlnDev* out = R.pa + pa_base + my_off;
                        double prev_prob = INFINITY;
                        uint32_t prev_ord = 0;
                        bool first = true;
                        for (uint32_t e = 0; e < cnt; e++) {
                            double bp = -INFINITY; uint32_t bo = NONE32;
                            AlnRef b1{0.0, 0, 0, NONE32, false}, b2 = b1;
                            // decreasing ln_prob, ties in push order (locs.rs:795)
                            pc.enumerate([&](double prob, uint32_t ord, const AlnRef& x1, const AlnRef& x2) {
                                const bool after = first || prob < prev_prob || (prob == prev_prob && ord > prev_ord);
                                if (after && (prob > bp || (prob == bp && ord < bo))) { bp = prob; bo = ord; b1 = x1; b2 = x2; }
                            });
                            PairAlnDev o;
                            o.ln_prob = bp * weight;
                            o.mid1 = b1.idx == NONE32 ? NONE32 : (b1.start + b1.end) / 2;   // Interval::middle, interv.rs:154-156
                            o.mid2 = b2.idx == NONE32 ? NONE32 : (b2.start + b2.end) / 2;
                            o.contig = static_cast<uint16_t>(c);
                            o.ix1 = b1.idx == NONE32 ? 0xFFFFu : static_cast<uint16_t>(b1.idx);
                            o.ix2 = b2.idx == NONE32 ? 0xFFFFu : static_cast<uint16_t>(b2.idx);
                            o._pad = 0;
                            out[e] = o;
                            prev_prob = bp; prev_ord = bo; first = false;
                        }
                    }
                }
            }
        }

        // rows of pairs that are not in AllAlignments::reads contribute 0.0 to every genotype score
        if (status != LCTY_READ_GOOD)
            for (uint32_t c = lane; c < A; c += WAVE) mrow[c] = 0.0;
        if (lane == 0) {
            R.status[p] = status;
            R.weight[p] = accepted ? weight : 0.0;
            R.unmapped_prob[p] = accepted ? unmapped_prob : 0.0;
            R.pa_off[p] = accepted ? pa_base : 0ull;
            R.pa_cnt[p] = accepted ? total_cnt : 0u;
            if (!accepted) { R.uniq_kmers[2 * p] = 0; R.uniq_kmers[2 * p + 1] = 0; }
        }
        __syncthreads();
    }
}

static size_t score_lds_bytes(uint32_t max_recs, uint32_t A) {
    size_t b = static_cast<size_t>(max_recs) * sizeof(RecLds) + static_cast<size_t>(2) * A * sizeof(uint32_t);
    b += static_cast<size_t>((max_recs + 1) & ~1u) * sizeof(uint16_t) + static_cast<size_t>(3) * A;
    return (b + 15) & ~static_cast<size_t>(15);
}

void launch_score_reads(lcty_reads* reads) {
    lcty_ctx* ctx = reads->ctx;
    const LocusView L = reads->locus->view();
    const ReadsView R = reads->view();
    const uint32_t max_recs = std::max<uint32_t>(reads->max_recs_per_pair, 1);
    const size_t lds = score_lds_bytes(max_recs, L.n_alleles);
    const size_t lds_max = 160 * 1024;
    if (lds > lds_max)
        fail(LCTY_ERR_UNSUPPORTED,
             "a read pair with %u records on %u alleles needs %zu B of LDS (> %zu): not supported by this build",
             max_recs, L.n_alleles, lds, lds_max);
    if (max_recs > 65535) fail(LCTY_ERR_UNSUPPORTED, "more than 65535 records in one read pair");
    if (lds > 48 * 1024)
        LCTY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(score_reads_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    const uint32_t cus = static_cast<uint32_t>(ctx->props.multiProcessorCount);
    const uint32_t per_cu = static_cast<uint32_t>(std::max<size_t>(1, std::min<size_t>(16, lds_max / lds)));
    const uint64_t grid = std::max<uint64_t>(1, std::min<uint64_t>(reads->n_pairs, static_cast<uint64_t>(cus) * per_cu));
    const char* dbg_env = getenv("LCTY_DBG");      // developer ablation switch; 0 / unset = the real kernel
    const uint32_t dbg = dbg_env ? static_cast<uint32_t>(atoi(dbg_env)) : 0u;
    reads->d_pa_count.zero(ctx->stream);
    ctx->timed(LCTY_K_SCORE, [&] {
        hipLaunchKernelGGL(score_reads_kernel, dim3(static_cast<uint32_t>(grid)), dim3(WAVE), lds, ctx->stream, L, R, max_recs, dbg);
    });
    LCTY_HIP(hipGetLastError());
}

}  // namespace lcty
