// lcty_locus.hip — per-locus preprocessing.
//   K1  UniqueKmers::new (src/model/locs.rs:930-963): device hash set of locus-unique canonical k-mers
//   K3  ContigInfo::new  (src/model/windows.rs:362-424): GC / unique-k-mer / complexity moving windows
//       (contig_info_kernel: thread per neighbourhood position, seen ck-mers as an LDS bit set)
//   LUTs: InsertDistr (bg/insertsz.rs:195-208), DistrCache (model/distr_cache.rs:61-75),
//         EditDistCache (bg/err_prof.rs:415-448)
#include <algorithm>
#include <atomic>
#include <memory>

#include "lcty_objects.hpp"

namespace lcty {

// ---------------------------------------------------------------------------------------------
// K1: hash-set build. One thread rolls over SEG consecutive k-mer start positions of one allele,
// restating kmers::kmers::<_, _, CANONICAL> (src/seq/kmers.rs:163-202) on the ASCII sequence.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t KSEG = 64;

__device__ inline void kset_insert(uint64_t* table, uint64_t mask, uint64_t key) {
    uint64_t slot = mix64(key) & mask;
    while (true) {
        const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&table[slot]),
                                                 static_cast<unsigned long long>(KSET_EMPTY),
                                                 static_cast<unsigned long long>(key));
        if (old == KSET_EMPTY || old == key) return;
        slot = (slot + 1) & mask;
    }
}

__global__ void kset_build_kernel(const uint8_t* __restrict__ seqs, const uint64_t* __restrict__ seq_off,
                                  const uint16_t* __restrict__ counts, const uint64_t* __restrict__ cnt_off,
                                  uint32_t k, uint64_t* table, uint64_t mask, uint32_t* undef_flag) {
    const uint32_t a = blockIdx.y;
    const uint8_t* seq = seqs + seq_off[a];
    const uint64_t len = seq_off[a + 1] - seq_off[a];
    const uint16_t* cnt = counts + cnt_off[a];
    if (len < k) return;
    const uint64_t nk = len + 1 - k;
    const uint64_t s = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * KSEG;
    if (s >= nk) return;
    const uint64_t e = min(s + KSEG, nk);
    const uint64_t kmask = (1ull << (2 * k)) - 1ull;
    const uint32_t rv_shift = 2 * k - 2;
    uint64_t fw = 0, rv = 0;
    uint64_t reset = s + k - 1;
    for (uint64_t i = s; i < e + k - 1; i++) {
        uint32_t enc;
        switch (seq[i]) {
            case 'A': enc = 0; break;
            case 'C': enc = 1; break;
            case 'G': enc = 2; break;
            case 'T': enc = 3; break;
            default: enc = 4;
        }
        if (enc == 4) {
            reset = i + k;
            if (i + 1 >= s + k && cnt[i + 1 - k] == 0) atomicOr(undef_flag, 1u);   // UNDEF enters the set (kmers.rs:184-190)
            continue;
        }
        fw = ((fw << 2) | enc) & kmask;
        rv = (rv >> 2) | (static_cast<uint64_t>(3 - enc) << rv_shift);
        if (i + 1 >= s + k) {
            const uint64_t pos = i + 1 - k;
            if (i >= reset) {
                if (cnt[pos] == 0) kset_insert(table, mask, rv < fw ? rv : fw);
            } else if (cnt[pos] == 0) {
                atomicOr(undef_flag, 1u);
            }
        }
    }
}

__global__ void kset_count_kernel(const uint64_t* __restrict__ table, uint64_t cap, unsigned long long* out) {
    uint64_t local = 0;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < cap;
         i += static_cast<uint64_t>(gridDim.x) * blockDim.x)
        local += table[i] != KSET_EMPTY;
    for (int off = WAVE / 2; off > 0; off >>= 1) local += __shfl_down(local, off);
    if ((threadIdx.x & (WAVE - 1)) == 0 && local) atomicAdd(out, static_cast<unsigned long long>(local));
}

__global__ void kset_compact_kernel(const uint64_t* __restrict__ big, uint64_t big_cap, uint64_t* small, uint64_t small_mask) {
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < big_cap;
         i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint64_t key = big[i];
        if (key != KSET_EMPTY) kset_insert(small, small_mask, key);
    }
}

// WeightCalculator::get (src/model/windows.rs:163-177) and ContigInfo::neighb_info (439-445) for every position
__device__ inline double weight_calc(double breakpoint, double power, double x) {
    const double const_fct = pow(breakpoint / (1.0 - breakpoint), power);
    return 1.0 / (1.0 + const_fct * pow((1.0 - x) / x, power));
}
__global__ void window_weight_kernel(const uint32_t* __restrict__ uniq_cnt, const uint16_t* __restrict__ compl_cnt, uint64_t n,
                                     double uniq_mult, double compl_mult, double kmers_bp, double kmers_pow,
                                     double compl_bp, double compl_pow, double* __restrict__ out) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double w = 1.0;
    if (kmers_bp > 0.0) w = weight_calc(kmers_bp, kmers_pow, static_cast<double>(uniq_cnt[i]) * uniq_mult);
    if (compl_bp > 0.0) w = w * weight_calc(compl_bp, compl_pow, static_cast<double>(compl_cnt[i]) * compl_mult);
    out[i] = w * 1.0;      // explicit weight == 1 (no --reg-weights file); explicit_window_kernel otherwise
}

// one factor of the window weight as a table over the count it is a function of: out[i] = what window_weight_kernel computes for a
// count of i (the same device code: the products of two entries are the kernel's weights bit for bit); out[n] = 0
__global__ void weight_table_kernel(uint32_t n, double mult, double bp, double pw, bool with_zero_tail, double* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = bp > 0.0 ? weight_calc(bp, pw, static_cast<double>(i) * mult) : 1.0;
    else if (i == n && with_zero_tail) out[i] = 0.0;
}

// NeighbInfo::explicit_weight of every moving-window position (windows.rs:409-413): the average of the window's own bases,
// taken from the running fixed-point sums as ExplicitWeights::average does (236-238: the INTEGER sum is divided by the
// number of bases, then scaled by 2^-32) and multiplied into the window weight as its last factor (441-443).
__global__ void explicit_window_kernel(const uint64_t* __restrict__ cum, const uint64_t* __restrict__ ew_off,
                                       const uint32_t* __restrict__ ci_off, uint32_t left_padding, uint32_t window,
                                       double* __restrict__ win_weight) {
    const uint32_t a = blockIdx.y;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t o = ci_off[a], n = ci_off[a + 1] - o;
    if (i >= n) return;
    const uint64_t* c = cum + ew_off[a];
    const uint32_t start = i + left_padding;
    const double avg = static_cast<double>((c[start + window] - c[start]) / static_cast<uint64_t>(window)) / 4294967296.0;
    win_weight[o + i] = win_weight[o + i] * avg;
}

static uint64_t next_pow2(uint64_t x) {
    uint64_t p = 1;
    while (p < x) p <<= 1;
    return p;
}

// ---------------------------------------------------------------------------------------------
// K3: ContigInfo::new (windows.rs:386-407) + linguistic_complexity (compl.rs:115-140), one thread per
// neighbourhood position. Everything is an integer numerator; the f64 values of the reference are count * mult.
//   gc        rounded percentage of C/G among the `neighb` bases (windows.rs:387-391)
//   uniq      k-mers with off-target count 0 among the neighb + 1 - k k-mers of the window (395-403)
//   compl     distinct (non-canonical) ck-mers among the neighb + 1 - ck of the window; every ck-mer that contains
//             a non-ACGT base is the single value UNDEF. Seen ck-mers are bits of a per-thread set in LDS
//             (word w of thread t at [w * 64 + t]).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void contig_info_kernel(const uint8_t* __restrict__ seqs, const uint64_t* __restrict__ seq_off,
                                                         const uint16_t* __restrict__ counts, const uint64_t* __restrict__ cnt_off,
                                                         const uint32_t* __restrict__ ci_off, uint32_t k, uint32_t neighb, uint32_t ck,
                                                         uint32_t words, uint8_t* __restrict__ gc, uint32_t* __restrict__ uniq,
                                                         uint16_t* __restrict__ compl_cnt) {
    extern __shared__ uint32_t seen[];                       // [words][64]
    const uint32_t a = blockIdx.y, tid = threadIdx.x;
    const uint8_t* seq = seqs + seq_off[a];
    const uint16_t* cnt = counts + cnt_off[a];
    const uint32_t len = static_cast<uint32_t>(seq_off[a + 1] - seq_off[a]);
    const uint32_t n_pos = len - neighb + 1;
    const uint32_t p = blockIdx.x * 64 + tid;
    if (blockIdx.x * 64 >= n_pos) return;
    const bool valid = p < n_pos;
    const uint32_t q = valid ? p : 0;                        // idle lanes shadow position 0
    for (uint32_t w = 0; w < words; w++) seen[w * 64 + tid] = 0;
    const uint32_t mask = ck >= 16 ? 0xFFFFFFFFu : ((1u << (2 * ck)) - 1u);
    uint32_t c_gc = 0, v = 0, bad = 0, undef_seen = 0;
    for (uint32_t i = 0; i < neighb; i++) {
        const uint8_t b = seq[q + i];
        c_gc += (b == 'C' || b == 'G');
        uint32_t enc;
        switch (b) { case 'A': enc = 0; break; case 'C': enc = 1; break; case 'G': enc = 2; break; case 'T': enc = 3; break;
                     default: enc = 4; }
        if (enc == 4) { bad = ck; v = (v << 2) & mask; } else { v = ((v << 2) | enc) & mask; if (bad) bad--; }
        if (i + 1 >= ck) {
            if (bad) undef_seen = 1;
            else seen[(v >> 5) * 64 + tid] |= 1u << (v & 31u);
        }
    }
    uint32_t distinct = undef_seen;
    for (uint32_t w = 0; w < words; w++) distinct += __popc(seen[w * 64 + tid]);
    uint32_t c_u = 0;
    const uint32_t span = neighb + 1 - k;
    for (uint32_t i = 0; i < span; i++) c_u += cnt[q + i] == 0;
    if (valid) {
        const uint64_t o = static_cast<uint64_t>(ci_off[a]) + p;
        gc[o] = static_cast<uint8_t>(round(100.0 / static_cast<double>(neighb) * static_cast<double>(c_gc)));
        uniq[o] = c_u;
        compl_cnt[o] = static_cast<uint16_t>(distinct);
    }
}

// The same for wide neighbourhoods (long reads: neighb = 2/3 of the read length): a thread takes CI_SEG consecutive positions, counts
// its first window as above and then SLIDES it — one base, one k-mer count and one ck-mer leave on the left, one of each enters on the
// right; the ck-mers of the window are counted (u16 per possible ck-mer and thread in LDS, [v * 64 + thread]), `distinct` follows
// the counts that reach or leave zero. O(neighb + CI_SEG) per thread instead of O(neighb) per position: the direct form took 119 ms
// for 256 alleles of 50 kb at the neighbourhood of 10-kb reads. Same integers.
constexpr uint32_t CI_SEG = 256;
__global__ __launch_bounds__(64) void contig_info_slide_kernel(const uint8_t* __restrict__ seqs, const uint64_t* __restrict__ seq_off,
                                                               const uint16_t* __restrict__ counts, const uint64_t* __restrict__ cnt_off,
                                                               const uint32_t* __restrict__ ci_off, uint32_t k, uint32_t neighb, uint32_t ck,
                                                               uint32_t n_kmers, uint8_t* __restrict__ gc, uint32_t* __restrict__ uniq,
                                                               uint16_t* __restrict__ compl_cnt) {
    extern __shared__ uint16_t held[];                       // [n_kmers][64]
    const uint32_t a = blockIdx.y, tid = threadIdx.x;
    const uint8_t* seq = seqs + seq_off[a];
    const uint16_t* cnt = counts + cnt_off[a];
    const uint32_t len = static_cast<uint32_t>(seq_off[a + 1] - seq_off[a]);
    const uint32_t n_pos = len - neighb + 1;
    const uint32_t p0 = (blockIdx.x * 64 + tid) * CI_SEG;
    if (blockIdx.x * 64 * CI_SEG >= n_pos) return;
    for (uint32_t v = 0; v < n_kmers; v++) held[v * 64 + tid] = 0;
    if (p0 >= n_pos) return;                                 // (no barrier below: every thread owns its column of `held`)
    const uint32_t p1 = min(n_pos, p0 + CI_SEG);
    const uint32_t mask = (1u << (2 * ck)) - 1u;
    auto enc_of = [](uint8_t b) -> uint32_t { return b == 'A' ? 0u : b == 'C' ? 1u : b == 'G' ? 2u : b == 'T' ? 3u : 4u; };
    uint32_t c_gc = 0, v = 0, bad = 0, undef_cnt = 0, distinct = 0;
    auto enter = [&](uint32_t value, bool is_undef) {
        if (is_undef) undef_cnt++;
        else { uint16_t& h = held[value * 64 + tid]; if (h++ == 0) distinct++; }
    };
    // the first window
    for (uint32_t i = 0; i < neighb; i++) {
        const uint8_t b = seq[p0 + i];
        c_gc += (b == 'C' || b == 'G');
        const uint32_t enc = enc_of(b);
        if (enc == 4) { bad = ck; v = (v << 2) & mask; } else { v = ((v << 2) | enc) & mask; if (bad) bad--; }
        if (i + 1 >= ck) enter(v, bad != 0);
    }
    const uint32_t span = neighb + 1 - k;
    uint32_t c_u = 0;
    for (uint32_t i = 0; i < span; i++) c_u += cnt[p0 + i] == 0;
    const double gc_mult = 100.0 / static_cast<double>(neighb);
    for (uint32_t p = p0;; p++) {
        const uint64_t o = static_cast<uint64_t>(ci_off[a]) + p;
        gc[o] = static_cast<uint8_t>(round(gc_mult * static_cast<double>(c_gc)));
        uniq[o] = c_u;
        compl_cnt[o] = static_cast<uint16_t>(distinct + (undef_cnt ? 1u : 0u));
        if (p + 1 >= p1) break;
        // slide to p + 1: position p leaves, position p + neighb enters
        const uint8_t out_b = seq[p], in_b = seq[p + neighb];
        c_gc += (in_b == 'C' || in_b == 'G');
        c_gc -= (out_b == 'C' || out_b == 'G');
        c_u += cnt[p + span] == 0;
        c_u -= cnt[p] == 0;
        // the ck-mer that started at p
        uint32_t lv = 0; bool lbad = false;
        for (uint32_t j = 0; j < ck; j++) { const uint32_t e = enc_of(seq[p + j]); lbad |= e == 4; lv = (lv << 2) | (e & 3u); }
        if (lbad) undef_cnt--;
        else { uint16_t& h = held[(lv & mask) * 64 + tid]; if (--h == 0) distinct--; }
        // the ck-mer that ends at p + neighb
        const uint32_t enc = enc_of(in_b);
        if (enc == 4) { bad = ck; v = (v << 2) & mask; } else { v = ((v << 2) | enc) & mask; if (bad) bad--; }
        enter(v, bad != 0);
    }
}

}  // namespace lcty

using namespace lcty;

LocusView lcty_locus::view() const {
    LocusView v{};
    v.n_alleles = n_alleles; v.k = k;
    v.allele_len = d_allele_len.p; v.ci_off = d_ci_off.p; v.compl_cnt = d_compl_cnt.p;
    v.compl_mult = compl_mult; v.half_neighb = half_neighb;
    v.kset = d_kset.p; v.kset_mask = kset_cap - 1; v.undef_in_set = undef_in_set;
    v.weight_mult = 1.0 / static_cast<double>(prm.kmer_soft_thresh + 1 - prm.kmer_hard_thresh);   // locs.rs:957
    v.weight_interc = (1.0 - static_cast<double>(prm.kmer_hard_thresh)) * v.weight_mult;           // locs.rs:958
    v.ins_lut = d_ins_lut.p; v.ins_lut_size = static_cast<uint32_t>(ins_lut.size());
    v.ins_n = ins.n; v.ins_lnq = ins.lnq; v.ins_lnpmf_const = ins.lnpmf_const; v.insert_penalty = insert_penalty;
    for (int i = 0; i < 5; i++) v.lp[i] = bg.op_lnprobs[i];
    v.edit_lut = d_edit_lut.p; v.edit_lut_size = edit_dev_size;
    v.unmapped_penalty = prm.unmapped_penalty; v.prob_diff = prm.prob_diff; v.min_weight = prm.min_weight;
    v.poor_compl = prm.poor_compl; v.poor_compl_edit = prm.poor_compl_edit;
    v.boundary = prm.boundary_size - static_cast<uint32_t>(prm.tweak);
    v.is_paired = bg.is_paired != 0; v.short_reads = bg.technology == LCTY_TECH_ILLUMINA;
    v.strict_subset = prm.strict_subset;
    v.ew_val = has_explicit ? d_ew_val.p : nullptr; v.ew_off = d_ew_off.p; v.half_window = bg.window / 2;
    return v;
}

void lcty_locus::ensure_edit_thresholds(const uint32_t* lens, size_t n) {
    std::lock_guard<std::mutex> lock(edit_mutex);
    uint32_t max_len = 0;
    for (size_t i = 0; i < n; i++) max_len = std::max(max_len, lens[i]);
    bool changed = false;
    if (edit_cache.size() < static_cast<size_t>(max_len) + 1) {
        edit_cache.resize(static_cast<size_t>(max_len) + 1, make_uint2(~0u, ~0u));
        changed = true;
    }
    for (size_t i = 0; i < n; i++) {
        uint2& e = edit_cache[lens[i]];
        if (e.x == ~0u && e.y == ~0u) {
            uint32_t g, p;
            math::edit_thresholds(bg, lens[i], &g, &p);
            e = make_uint2(g, p);
            changed = true;
        }
    }
    if (changed) {
        ctx->activate();
        LCTY_HIP(hipStreamSynchronize(ctx->stream));     // the old table may still be read by a running kernel
        d_edit_lut.alloc(edit_cache.size());
        d_edit_lut.upload(edit_cache.data(), edit_cache.size(), ctx->stream);
        LCTY_HIP(hipStreamSynchronize(ctx->stream));
        edit_dev_size = static_cast<uint32_t>(edit_cache.size());
    }
}

extern "C" {

int32_t lcty_locus_create(lcty_ctx* ctx, uint32_t n_alleles, const uint8_t* seqs, const uint64_t* seq_off,
                          const uint16_t* offtarget, const uint64_t* cnt_off, uint32_t k,
                          const lcty_bg* bg, const lcty_params* params, lcty_locus** out) {
    return guarded([&] {
        if (!ctx || !seqs || !seq_off || !offtarget || !cnt_off || !bg || !params || !out)
            fail(LCTY_ERR_INVALID_INPUT, "lcty_locus_create: null argument");
        if (n_alleles == 0 || n_alleles > 65535)
            fail(LCTY_ERR_INVALID_INPUT, "number of alleles (%u) must be in 1..65535 (seq/contigs.rs:96-98)", n_alleles);
        if (k < 2) fail(LCTY_ERR_INVALID_INPUT, "k-mer size (%u) must be over 1 (locs.rs:937)", k);
        if (k > 63) fail(LCTY_ERR_INVALID_INPUT, "k-mer size (%u) must be at most 63 (kmers.rs:24-26: u128 k-mers)", k);
        if (params->tweak < 0 || std::isnan(params->prob_diff) || std::isnan(params->unmapped_penalty))
            fail(LCTY_ERR_INVALID_INPUT, "params have unresolved auto fields: call lcty_params_resolve first");
        if (static_cast<uint32_t>(params->tweak) >= params->boundary_size)
            fail(LCTY_ERR_INVALID_INPUT, "boundary size (%u) must be greater than tweak size (%d)", params->boundary_size, params->tweak);
        if (params->kmer_hard_thresh > params->kmer_soft_thresh)
            fail(LCTY_ERR_INVALID_INPUT, "hard k-mer threshold must not exceed the soft threshold");
        if (params->complexity_k == 0 || params->complexity_k > 15)
            fail(LCTY_ERR_INVALID_INPUT, "complexity k (%u) must be in 1..15", params->complexity_k);
        if (params->n_alt_cn > LCTY_MAX_ALT_CN) fail(LCTY_ERR_INVALID_INPUT, "too many alternative copy numbers");
        ctx->activate();

        auto L = std::unique_ptr<lcty_locus>(new lcty_locus());
        static std::atomic<uint64_t> next_serial{1};
        L->serial = next_serial.fetch_add(1);
        L->ctx = ctx; L->n_alleles = n_alleles; L->k = k; L->bg = *bg; L->prm = *params;
        const uint32_t neighb = bg->neighb, window = bg->window, ck = params->complexity_k;
        if (window == 0 || neighb < window) fail(LCTY_ERR_INVALID_DATA, "bg_depth: neighbourhood (%u) < window (%u)", neighb, window);
        if (neighb + 1 <= k || neighb + 1 <= ck) fail(LCTY_ERR_INVALID_DATA, "neighbourhood size (%u) too small for k = %u", neighb, k);
        const uint32_t boundary = params->boundary_size - static_cast<uint32_t>(params->tweak);

        L->allele_len.resize(n_alleles);
        L->ci_off.resize(n_alleles + 1);
        L->n_windows.resize(n_alleles);
        L->reg_start.resize(n_alleles);
        uint64_t total_pos = 0;
        for (uint32_t a = 0; a < n_alleles; a++) {
            const uint64_t len64 = seq_off[a + 1] - seq_off[a];
            if (len64 >= (1ull << 32)) fail(LCTY_ERR_INVALID_DATA, "allele %u is too long", a);
            const uint32_t len = static_cast<uint32_t>(len64);
            if (len < window + 2 * params->boundary_size)
                fail(LCTY_ERR_RUNTIME, "Contig %u is too short (len = %u)", a, len);                       // windows.rs:375-378
            if (!(len > 2 * boundary)) fail(LCTY_ERR_RUNTIME, "Some contigs are too short (must be over twice boundary size = %u)", 2 * boundary);
            if (len < neighb) fail(LCTY_ERR_RUNTIME, "Contig %u is shorter than the neighbourhood size", a);
            if (cnt_off[a + 1] - cnt_off[a] != len64 + 1 - k)
                fail(LCTY_ERR_INVALID_DATA, "k-mer counts of allele %u do not match its length (locs.rs:944)", a);
            L->allele_len[a] = len;
            L->ci_off[a] = static_cast<uint32_t>(total_pos);
            total_pos += len - neighb + 1;
            if (total_pos >= (1ull << 32)) fail(LCTY_ERR_UNSUPPORTED, "locus too large for 32-bit window offsets");
            L->n_windows[a] = (len - 2 * params->boundary_size) / window;                                   // windows.rs:380
            L->reg_start[a] = (len - L->n_windows[a] * window) / 2;                                         // windows.rs:381-382
        }
        L->ci_off[n_alleles] = static_cast<uint32_t>(total_pos);
        L->left_padding = (neighb - window) / 2;
        L->half_neighb = neighb / 2;
        L->uniq_mult = 1.0 / static_cast<double>(neighb + 1 - k);
        L->compl_mult = 1.0 / static_cast<double>(std::min<uint64_t>(neighb + 1 - ck, 1ull << (2 * ck)));

        hipStream_t s = ctx->stream;
        DevBuf<uint8_t>& d_seqs = L->d_seqs;                              // stay resident: alignment recovery reads them
        DevBuf<uint64_t>& d_seq_off = L->d_seq_off;
        DevBuf<uint64_t> d_cnt_off; DevBuf<uint16_t> d_counts;
        const uint64_t total_seq = seq_off[n_alleles], total_cnt = cnt_off[n_alleles];
        d_seqs.alloc(total_seq); d_seqs.upload(seqs, total_seq, s);
        d_seq_off.alloc(n_alleles + 1); d_seq_off.upload(seq_off, n_alleles + 1, s);
        d_cnt_off.alloc(n_alleles + 1); d_cnt_off.upload(cnt_off, n_alleles + 1, s);
        d_counts.alloc(total_cnt); d_counts.upload(offtarget, total_cnt, s);
        L->d_allele_len.alloc(n_alleles); L->d_allele_len.upload(L->allele_len.data(), n_alleles, s);
        L->d_ci_off.alloc(n_alleles + 1); L->d_ci_off.upload(L->ci_off.data(), n_alleles + 1, s);

        // ---- K3 (device) ----
        L->d_compl_cnt.alloc(total_pos); L->d_gc.alloc(total_pos); L->d_uniq_cnt.alloc(total_pos);
        {
            if (ck > 7) fail(LCTY_ERR_UNSUPPORTED, "complexity k-mer size %u: the device handles up to 7", ck);
            const uint32_t words = std::max(1u, (1u << (2 * ck)) / 32u);
            const size_t lds = static_cast<size_t>(words) * 64 * sizeof(uint32_t);
            if (lds > 48 * 1024)
                LCTY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(contig_info_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(lds)));
            const uint32_t max_len0 = *std::max_element(L->allele_len.begin(), L->allele_len.end());
            const uint32_t n_ckmers = 1u << (2 * ck);
            const size_t lds_slide = static_cast<size_t>(n_ckmers) * 64 * sizeof(uint16_t);
            // wide neighbourhoods (long reads) slide their window; lcty_ctx_set_knob "contig_info_slide" 0 / 1 forces either form
            const int64_t want = ctx->knob("contig_info_slide", -1);
            const bool slide = lds_slide <= 144 * 1024 && neighb <= 65535 && (want >= 0 ? want != 0 : neighb >= 1024);
            if (slide) {
                if (lds_slide > 48 * 1024)
                    LCTY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(contig_info_slide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(lds_slide)));
                const uint32_t n_pos_max = max_len0 - neighb + 1;
                const dim3 grid((n_pos_max + 64 * CI_SEG - 1) / (64 * CI_SEG), n_alleles);
                hipLaunchKernelGGL(contig_info_slide_kernel, grid, dim3(64), lds_slide, s, d_seqs.p, d_seq_off.p, d_counts.p, d_cnt_off.p, L->d_ci_off.p,
                                   k, neighb, ck, n_ckmers, L->d_gc.p, L->d_uniq_cnt.p, L->d_compl_cnt.p);
            } else {
                const dim3 grid((max_len0 - neighb + 1 + 63) / 64, n_alleles);
                hipLaunchKernelGGL(contig_info_kernel, grid, dim3(64), lds, s, d_seqs.p, d_seq_off.p, d_counts.p, d_cnt_off.p, L->d_ci_off.p,
                                   k, neighb, ck, words, L->d_gc.p, L->d_uniq_cnt.p, L->d_compl_cnt.p);
            }
            LCTY_HIP(hipGetLastError());
        }
        L->d_n_windows.alloc(n_alleles); L->d_n_windows.upload(L->n_windows.data(), n_alleles, s);
        L->d_reg_start.alloc(n_alleles); L->d_reg_start.upload(L->reg_start.data(), n_alleles, s);
        L->max_n_windows = *std::max_element(L->n_windows.begin(), L->n_windows.end());
        L->d_win_weight.alloc(total_pos);
        hipLaunchKernelGGL(window_weight_kernel, dim3(static_cast<uint32_t>((total_pos + 255) / 256)), dim3(256), 0, s,
                           L->d_uniq_cnt.p, L->d_compl_cnt.p, total_pos, L->uniq_mult, L->compl_mult,
                           params->kmers_weight_bp, params->kmers_weight_pow, params->compl_weight_bp, params->compl_weight_pow,
                           L->d_win_weight.p);
        LCTY_HIP(hipGetLastError());
        {
            // counts run 0..neighb + 1 - k and 0..min(neighb + 1 - ck, 4^ck)
            const uint64_t n_u = static_cast<uint64_t>(neighb) + 1 - k + 1, n_c = std::min<uint64_t>(neighb + 1 - ck, 1ull << (2 * ck)) + 1;
            L->weight_tables_valid = n_u + 1 <= 512 && n_c <= 512;
            if (L->weight_tables_valid) {
                L->d_wk.alloc(n_u + 1); L->d_wc.alloc(n_c);
                hipLaunchKernelGGL(weight_table_kernel, dim3(3), dim3(256), 0, s, static_cast<uint32_t>(n_u), L->uniq_mult,
                                   params->kmers_weight_bp, params->kmers_weight_pow, true, L->d_wk.p);
                hipLaunchKernelGGL(weight_table_kernel, dim3(3), dim3(256), 0, s, static_cast<uint32_t>(n_c), L->compl_mult,
                                   params->compl_weight_bp, params->compl_weight_pow, false, L->d_wc.p);
                LCTY_HIP(hipGetLastError());
            }
        }

        // ---- K1: k-mers of 32..63 bases on the host (128-bit keys; the walk of kset_build_kernel) ----
        if (k > 31) {
            typedef unsigned __int128 u128;
            const u128 kmask = (static_cast<u128>(1) << (2 * k)) - 1;
            const uint32_t rv_shift = 2 * k - 2;
            std::vector<u128> keys;
            bool undef = false;
            for (uint32_t a = 0; a < n_alleles; a++) {
                const uint8_t* seq = seqs + seq_off[a];
                const uint64_t len = seq_off[a + 1] - seq_off[a];
                const uint16_t* cnt = offtarget + cnt_off[a];
                if (len < k) continue;
                u128 fw = 0, rv = 0;
                uint64_t reset = k - 1;
                for (uint64_t i = 0; i < len; i++) {
                    const uint8_t ch = seq[i];
                    const uint32_t enc = ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 4u;
                    if (enc == 4) {
                        reset = i + k;
                        if (i + 1 >= k && cnt[i + 1 - k] == 0) undef = true;         // UNDEF enters the set (kmers.rs:184-190)
                        continue;
                    }
                    fw = ((fw << 2) | enc) & kmask;
                    rv = (rv >> 2) | (static_cast<u128>(3 - enc) << rv_shift);
                    if (i + 1 >= k) {
                        const uint64_t pos = i + 1 - k;
                        if (cnt[pos] != 0) continue;
                        if (i >= reset) keys.push_back(rv < fw ? rv : fw); else undef = true;
                    }
                }
            }
            std::sort(keys.begin(), keys.end());
            keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
            L->undef_in_set = undef ? 1u : 0u;
            L->n_unique = keys.size() + (undef ? 1 : 0);
            L->kset_cap = next_pow2(std::max<uint64_t>(4 * keys.size(), 1024));
            std::vector<uint64_t> table(2 * L->kset_cap, KSET_EMPTY);
            for (const u128 key : keys) {
                const uint64_t lo = static_cast<uint64_t>(key), hi = static_cast<uint64_t>(key >> 64);
                uint64_t slot = kmer128_hash(lo, hi) & (L->kset_cap - 1);
                while (table[2 * slot] != KSET_EMPTY || table[2 * slot + 1] != KSET_EMPTY) slot = (slot + 1) & (L->kset_cap - 1);
                table[2 * slot] = lo; table[2 * slot + 1] = hi;
            }
            L->d_kset.alloc(table.size()); L->d_kset.upload(table.data(), table.size(), s);
            LCTY_HIP(hipStreamSynchronize(s));
        } else
        // ---- K1 (device) ----
        {
            DevBuf<uint64_t> d_big; DevBuf<uint32_t> d_flag; DevBuf<unsigned long long> d_n;
            const uint64_t big_cap = next_pow2(std::max<uint64_t>(2 * total_cnt, 1024));      // at most one key per k-mer position
            d_big.alloc(big_cap);
            LCTY_HIP(hipMemsetAsync(d_big.p, 0xFF, big_cap * sizeof(uint64_t), s));
            d_flag.alloc(1); d_flag.zero(s);
            d_n.alloc(1); d_n.zero(s);
            uint32_t max_len = *std::max_element(L->allele_len.begin(), L->allele_len.end());
            const uint32_t threads = 128;
            const uint32_t segs = (max_len + KSEG - 1) / KSEG;
            dim3 grid((segs + threads - 1) / threads, n_alleles);
            hipLaunchKernelGGL(kset_build_kernel, grid, dim3(threads), 0, s, d_seqs.p, d_seq_off.p, d_counts.p,
                               d_cnt_off.p, k, d_big.p, big_cap - 1, d_flag.p);
            hipLaunchKernelGGL(kset_count_kernel, dim3(1024), dim3(256), 0, s, d_big.p, big_cap, d_n.p);
            unsigned long long n_distinct = 0; uint32_t flag = 0;
            LCTY_HIP(hipMemcpyAsync(&n_distinct, d_n.p, sizeof(n_distinct), hipMemcpyDeviceToHost, s));
            LCTY_HIP(hipMemcpyAsync(&flag, d_flag.p, sizeof(flag), hipMemcpyDeviceToHost, s));
            LCTY_HIP(hipStreamSynchronize(s));
            L->undef_in_set = flag;
            L->n_unique = n_distinct + (flag ? 1 : 0);
            L->kset_cap = next_pow2(std::max<uint64_t>(4 * n_distinct, 1024));
            L->d_kset.alloc(L->kset_cap);
            LCTY_HIP(hipMemsetAsync(L->d_kset.p, 0xFF, L->kset_cap * sizeof(uint64_t), s));
            hipLaunchKernelGGL(kset_compact_kernel, dim3(1024), dim3(256), 0, s, d_big.p, big_cap, L->d_kset.p, L->kset_cap - 1);
            LCTY_HIP(hipGetLastError());
            LCTY_HIP(hipStreamSynchronize(s));
        }

        // ---- LUTs ----
        if (bg->is_paired) {
            if (!(bg->ins_n > 0.0 && bg->ins_p >= 0.0 && bg->ins_p <= 1.0))
                fail(LCTY_ERR_INVALID_DATA, "Incorrect Negative Binomial parameters n = %g, p = %g", bg->ins_n, bg->ins_p);
            L->ins = math::NBinom(bg->ins_n, bg->ins_p);
            const size_t sz = math::insert_cache_size(L->ins);
            L->ins_lut.resize(sz);
            for (size_t i = 0; i < sz; i++) L->ins_lut[i] = L->ins.ln_pmf(static_cast<uint32_t>(i));
            L->insert_penalty = L->ins.ln_pmf(L->ins.mode());
        } else {
            L->insert_penalty = std::numeric_limits<double>::quiet_NaN();
        }
        L->d_ins_lut.alloc(std::max<size_t>(L->ins_lut.size(), 1));
        if (!L->ins_lut.empty()) L->d_ins_lut.upload(L->ins_lut.data(), L->ins_lut.size(), s);
        {
            std::vector<double> lut(static_cast<size_t>(LCTY_GC_BINS) * LCTY_DEPTH_CACHE);
            for (uint32_t gc = 0; gc < LCTY_GC_BINS; gc++) {
                if (!(bg->depth_n[gc] > 0.0)) fail(LCTY_ERR_INVALID_DATA, "bg_depth: invalid NBinom at GC %u", gc);
                math::DepthDistr dd(*bg, *params, gc);
                for (uint32_t d = 0; d < LCTY_DEPTH_CACHE; d++) lut[gc * LCTY_DEPTH_CACHE + d] = dd.ln_pmf(d);
            }
            L->d_depth_lut.alloc(lut.size());
            L->d_depth_lut.upload(lut.data(), lut.size(), s);
            std::vector<DepthNB> nbs(LCTY_GC_BINS);
            for (uint32_t gc = 0; gc < LCTY_GC_BINS; gc++) {
                math::DepthDistr dd(*bg, *params, gc);
                DepthNB& d = nbs[gc];
                memset(&d, 0, sizeof(d));
                d.lnq = dd.cn1.lnq; d.n[0] = dd.cn1.n; d.lnpmf_const[0] = dd.cn1.lnpmf_const;
                for (size_t i = 0; i < dd.alts.size(); i++) { d.n[i + 1] = dd.alts[i].n; d.lnpmf_const[i + 1] = dd.alts[i].lnpmf_const; }
            }
            L->d_depth_nb.alloc(LCTY_GC_BINS);
            L->d_depth_nb.upload(nbs.data(), LCTY_GC_BINS, s);
            LCTY_HIP(hipStreamSynchronize(s));
        }
        LCTY_HIP(hipStreamSynchronize(s));
        *out = L.release();
    });
}

void lcty_locus_destroy(lcty_locus* locus) {
    if (!locus) return;
    (void)hipSetDevice(locus->ctx->device);
    delete locus;
}

int32_t lcty_locus_n_unique_kmers(const lcty_locus* locus, uint64_t* out) {
    return guarded([&] {
        if (!locus || !out) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        *out = locus->n_unique;
    });
}

int32_t lcty_locus_contig_info(const lcty_locus* locus, uint32_t allele, uint8_t* gc, uint32_t* uniq_cnt,
                               uint16_t* compl_cnt, uint32_t* n_windows, uint32_t* reg_start) {
    return guarded([&] {
        if (!locus) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (allele >= locus->n_alleles) fail(LCTY_ERR_INVALID_INPUT, "allele index out of range");
        const size_t o = locus->ci_off[allele], n = locus->ci_off[allele + 1] - o;
        hipStream_t s = locus->ctx->stream;
        locus->ctx->activate();
        if (gc) locus->d_gc.download(gc, n, s, o);
        if (uniq_cnt) locus->d_uniq_cnt.download(uniq_cnt, n, s, o);
        if (compl_cnt) locus->d_compl_cnt.download(compl_cnt, n, s, o);
        LCTY_HIP(hipStreamSynchronize(s));
        if (n_windows) *n_windows = locus->n_windows[allele];
        if (reg_start) *reg_start = locus->reg_start[allele];
    });
}

int32_t lcty_locus_edit_thresholds(const lcty_locus* locus, uint32_t read_len, uint32_t* good, uint32_t* passable) {
    return guarded([&] {
        if (!locus || !good || !passable) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        math::edit_thresholds(locus->bg, read_len, good, passable);
    });
}

int32_t lcty_locus_insert_lnprob(const lcty_locus* locus, uint32_t n, const uint32_t* sizes, double* out, double* insert_penalty) {
    return guarded([&] {
        if (!locus) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (!locus->bg.is_paired) fail(LCTY_ERR_INVALID_INPUT, "insert size distribution is undefined for single-end data");
        for (uint32_t i = 0; i < n; i++)
            out[i] = sizes[i] < locus->ins_lut.size() ? locus->ins_lut[sizes[i]] : locus->ins.ln_pmf(sizes[i]);
        if (insert_penalty) *insert_penalty = locus->insert_penalty;
    });
}

int32_t lcty_locus_set_explicit_weights(lcty_locus* locus, uint32_t n_lines, const uint32_t* allele, const uint32_t* start,
                                        const uint32_t* end, const double* value) {
    return guarded([&] {
        if (!locus || (n_lines && (!allele || !start || !end || !value))) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        lcty_locus* L = locus;
        const uint32_t A = L->n_alleles;
        // load_explicit_weights (windows.rs:257-317) on parsed lines: per allele len + 1 (value, running sum) entries
        std::vector<uint64_t> off(A + 1, 0);
        for (uint32_t a = 0; a < A; a++) off[a + 1] = off[a] + L->allele_len[a] + 1;
        std::vector<double> val(off[A]);
        std::vector<uint64_t> cum(off[A]);
        std::vector<uint32_t> filled(A, 0);
        std::vector<uint64_t> sum(A, 0);
        for (uint32_t t = 0; t < n_lines; t++) {
            const uint32_t a = allele[t];
            if (a >= A) continue;                                             // unknown contig: the line is skipped (269-272)
            if (start[t] >= end[t] || end[t] > L->allele_len[a])
                fail(LCTY_ERR_INVALID_INPUT, "explicit weights, line %u: interval %u-%u out of range on allele %u (length %u)",
                     t, start[t], end[t], a, L->allele_len[a]);              // interv.rs:112-116
            if (!(value[t] >= 0.0 && value[t] <= 1.0))
                fail(LCTY_ERR_INVALID_DATA, "Failed to parse explicit weights (line %u): value must be in [0, 1]", t);
            if (filled[a] != start[t])
                fail(LCTY_ERR_INVALID_DATA, "Failed to parse explicit weights: haplotype %u not fully covered", a);
            const uint64_t inc = static_cast<uint64_t>(value[t] * 4294967296.0);                      // extend_by, 212-218
            double* v = val.data() + off[a];
            uint64_t* c = cum.data() + off[a];
            for (uint32_t i = start[t]; i < end[t]; i++) { v[i] = value[t]; c[i] = sum[a]; sum[a] += inc; }
            filled[a] = end[t];
        }
        for (uint32_t a = 0; a < A; a++) {
            if (filled[a] == 0) fail(LCTY_ERR_INVALID_DATA, "Failed to parse explicit weights: haplotype %u missing", a);
            if (filled[a] != L->allele_len[a])
                fail(LCTY_ERR_INVALID_DATA, "Failed to parse explicit weights: haplotype %u not fully covered/has different length", a);
            val[off[a] + filled[a]] = val[off[a] + filled[a] - 1];                                    // finish(), 221-224
            cum[off[a] + filled[a]] = sum[a];
        }
        L->ctx->activate();
        hipStream_t s = L->ctx->stream;
        LCTY_HIP(hipStreamSynchronize(s));                  // a kernel still reading the old tables
        DevBuf<uint64_t> d_cum;
        d_cum.alloc(off[A]); d_cum.upload(cum.data(), off[A], s);
        L->d_ew_val.alloc(off[A]); L->d_ew_val.upload(val.data(), off[A], s);
        L->d_ew_off.alloc(A + 1); L->d_ew_off.upload(off.data(), A + 1, s);
        // window weights from scratch (a second call replaces the first), then the window averages as the last factor
        const uint64_t total_pos = L->ci_off[A];
        hipLaunchKernelGGL(window_weight_kernel, dim3(static_cast<uint32_t>((total_pos + 255) / 256)), dim3(256), 0, s,
                           L->d_uniq_cnt.p, L->d_compl_cnt.p, total_pos, L->uniq_mult, L->compl_mult,
                           L->prm.kmers_weight_bp, L->prm.kmers_weight_pow, L->prm.compl_weight_bp, L->prm.compl_weight_pow,
                           L->d_win_weight.p);
        uint32_t max_pos = 0;
        for (uint32_t a = 0; a < A; a++) max_pos = std::max(max_pos, L->ci_off[a + 1] - L->ci_off[a]);
        hipLaunchKernelGGL(explicit_window_kernel, dim3((max_pos + 255) / 256, A), dim3(256), 0, s, d_cum.p, L->d_ew_off.p,
                           L->d_ci_off.p, L->left_padding, L->bg.window, L->d_win_weight.p);
        LCTY_HIP(hipGetLastError());
        LCTY_HIP(hipStreamSynchronize(s));                  // host vectors and d_cum go out of scope
        L->has_explicit = true;
    });
}

int32_t lcty_locus_window_weights(const lcty_locus* locus, double* out) {
    return guarded([&] {
        if (!locus || !out) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        locus->ctx->activate();
        locus->d_win_weight.download(out, locus->d_win_weight.n, locus->ctx->stream);
        LCTY_HIP(hipStreamSynchronize(locus->ctx->stream));
    });
}

int32_t lcty_locus_depth_lut(const lcty_locus* locus, double* out) {
    return guarded([&] {
        if (!locus || !out) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        locus->ctx->activate();
        locus->d_depth_lut.download(out, static_cast<size_t>(LCTY_GC_BINS) * LCTY_DEPTH_CACHE, locus->ctx->stream);
        LCTY_HIP(hipStreamSynchronize(locus->ctx->stream));
    });
}

}  // extern "C"
