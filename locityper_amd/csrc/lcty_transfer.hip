// lcty_transfer.hip — alignment recovery of AllAlignments::load (K6):
//   HapAlns::{add, sort_best_ixs, transfer_alignments}   src/seq/transfer.rs:21-140
//   CigarIndex                                            src/seq/cigar.rs:1085-1162
//   Cigar::transfer_read_alignment + the aligner           lcty_transfer_device.hpp
//
// Recovery runs between two scoring passes. The first pass tells which read pairs reach recover_and_group_alignments
// (locs.rs:1255-1260: well mapped, in bounds, weight >= min_weight). For those, `transfer_kernel` (one wavefront per read
// pair) rebuilds the PrelimAlignments of the pair (push order, 128-bp position set, thresholds: locs.rs:298-344, 502-567),
// walks its alignments in order as the reference does, and gives every (alignment, target haplotype) transfer of a source
// to its own lane. Transferred alignments are appended to the batch's record table — after the original records of their
// read end, which is the order PrelimAlignments::push sees them in within an end — and the second scoring pass treats them
// like any other record.
#include <algorithm>
#include <memory>
#include <numeric>
#include <system_error>
#include <thread>

#include "lcty_objects.hpp"
#include "lcty_transfer_device.hpp"

namespace lcty {

using namespace xfer;

constexpr uint32_t NONE32T = 0xFFFFFFFFu;

struct HapView {
    uint32_t A, transfer_fails;
    const uint32_t* cell_of;       // [A * A] at (min, max): cell index or NONE
    const uint32_t* item_off;      // [cells + 1]
    const uint2* items;            // {op, len}, query = lower contig id
    const uint2* positions;        // {qpos, rpos} at the start of every item
    const uint32_t* sparse_off;    // [2 * cells + 1]: direction 0 (query -> ref) then 1 of every cell
    const uint2* sparse;           // {cigar_ix, pos}
    const uint32_t* best_off;      // [A + 1]
    const uint32_t* best_ids;
    const uint4* best_meta;        // [best entries] {target, first item of the pair's alignment, number of items, first sparse entry of the direction}
    const uint8_t* seqs; const uint64_t* seq_off;
};

// one scored alignment of the pair (Alignment + what push needs)
struct PAln {
    double ln_prob;
    uint32_t start, contig_end;    // contig | read end << 16 | reverse << 17
    uint32_t edit, src;            // src: record index, or 0x80000000 | index of a transferred alignment
};
struct NewAln {                    // a transferred alignment, in push order
    double ln_prob;
    uint32_t start, contig_end, edit, n_cigar, cigar_at;     // cigar_at: word offset in the pair's new-CIGAR block
    uint32_t pushed;
};

struct PairScratch {
    PAln* alns; uint32_t cap_alns;
    uint64_t* hkey; uint2* hval; uint32_t hcap;         // position set: key -> {index, pos}; key 0 = free (real keys have bits 48+ set)
    uint8_t* seen;
    uint4* pre;                                          // [cap_alns] scores of the input records with long CIGARs, made by the whole wavefront
    NewAln* news; uint32_t cap_new;
    uint32_t* words; uint32_t cap_words;
    uint8_t* lanes;                                      // 64 x lane_scratch_bytes
};

__host__ __device__ inline size_t pair_scratch_bytes(uint32_t cap_alns, uint32_t hcap, uint32_t cap_new, uint32_t cap_words, const Limits& lim) {
    size_t b = 0;
    b += sizeof(PAln) * cap_alns; b = (b + 15) & ~size_t(15);
    b += 8 * hcap + 8 * hcap;
    b += (cap_alns + 15) & ~15u;
    b += sizeof(uint4) * static_cast<size_t>(cap_alns);
    b += sizeof(NewAln) * cap_new; b = (b + 15) & ~size_t(15);
    b += 4 * static_cast<size_t>(cap_words); b = (b + 127) & ~size_t(127);            // the lanes' chunks of 64 bytes lie inside cache lines
    b += 64 * lane_scratch_bytes(lim);
    return (b + 255) & ~size_t(255);
}

__device__ inline PairScratch carve(uint8_t* base, uint32_t cap_alns, uint32_t hcap, uint32_t cap_new, uint32_t cap_words) {
    PairScratch s;
    size_t b = 0;
    s.alns = reinterpret_cast<PAln*>(base); s.cap_alns = cap_alns; b += sizeof(PAln) * cap_alns; b = (b + 15) & ~size_t(15);
    s.hkey = reinterpret_cast<uint64_t*>(base + b); b += 8 * hcap;
    s.hval = reinterpret_cast<uint2*>(base + b); b += 8 * hcap; s.hcap = hcap;
    s.seen = base + b; b += (cap_alns + 15) & ~15u;
    s.pre = reinterpret_cast<uint4*>(base + b); b += sizeof(uint4) * static_cast<size_t>(cap_alns);
    s.news = reinterpret_cast<NewAln*>(base + b); s.cap_new = cap_new; b += sizeof(NewAln) * cap_new; b = (b + 15) & ~size_t(15);
    s.words = reinterpret_cast<uint32_t*>(base + b); s.cap_words = cap_words; b += 4 * static_cast<size_t>(cap_words); b = (b + 127) & ~size_t(127);
    s.lanes = base + b;
    return s;
}

// count_region_operations_fast (aln.rs:288-317) + ErrorProfile::ln_prob (err_prof.rs:212-221); word(i) = raw CIGAR word i
struct Scored { double ln_prob; uint32_t start, end, edit; bool bad; };
template <typename WordAt>
__device__ inline Scored score_cigar(const LocusView& L, WordAt word, uint32_t n, uint32_t pos, uint32_t contig_len, bool primary) {
    uint32_t matches = 0, mism = 0, ins = 0, del = 0, left = 0, right = 0;
    bool bad = false;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t raw = word(i);
        uint32_t op = raw & 15u;
        const uint32_t len = raw >> 4;
        const bool edge = i == 0 || i + 1 == n;
        if (op == OP_H) { if (edge && !primary) op = OP_S; else bad = true; }
        if (op == OP_EQ) matches += len; else if (op == OP_X) mism += len; else if (op == OP_I) ins += len; else if (op == OP_D) del += len;
        else if (op == OP_S) { if (i == 0) left = len; if (i + 1 == n) right = len; }
        else bad = true;
    }
    Scored s;
    const uint32_t ref_len = matches + mism + del;
    s.start = pos; s.end = pos + ref_len;
    const uint32_t clip = min(left, pos) + min(right, contig_len > s.end ? contig_len - s.end : 0u);
    const uint32_t common = mism + ins + clip;
    s.edit = common + del;
    s.ln_prob = L.lp[0] * static_cast<double>(matches) + L.lp[1] * static_cast<double>(mism) + L.lp[2] * static_cast<double>(ins)
              + L.lp[3] * static_cast<double>(del) + L.lp[4] * static_cast<double>(clip);
    s.bad = bad;
    return s;
}
__device__ inline Scored score_words(const LocusView& L, const uint32_t* raw, uint32_t n, uint32_t pos, uint32_t contig_len, bool primary) {
    return score_cigar(L, [raw](uint32_t i) { return raw[i]; }, n, pos, contig_len, primary);
}
// The same by all 64 lanes (a 10-kb read's CIGAR is ~3 000 words): lane l counts the words l, l + 64, ..., the counts are summed across
// the lanes (integers: the order does not matter) and every lane ends with the result score_words gives.
constexpr uint32_t WAVE_SCORE_FROM = 32;          // words from which a record's CIGAR is counted by the wavefront
__device__ inline Scored score_words_wave(const LocusView& L, const uint32_t* raw, uint32_t n, uint32_t pos, uint32_t contig_len, bool primary, uint32_t lane) {
    uint32_t c[6] = {0, 0, 0, 0, 0, 0};                  // matches, mismatches, insertions, deletions, left clip, right clip
    bool bad = false;
    for (uint32_t i = lane; i < n; i += 64) {
        const uint32_t w = raw[i];
        uint32_t op = w & 15u;
        const uint32_t len = w >> 4;
        const bool edge = i == 0 || i + 1 == n;
        if (op == OP_H) { if (edge && !primary) op = OP_S; else bad = true; }
        if (op == OP_EQ) c[0] += len; else if (op == OP_X) c[1] += len; else if (op == OP_I) c[2] += len; else if (op == OP_D) c[3] += len;
        else if (op == OP_S) { if (i == 0) c[4] = len; if (i + 1 == n) c[5] = len; }
        else bad = true;
    }
#pragma unroll
    for (int k = 0; k < 6; k++)
        for (int o = 32; o > 0; o >>= 1) c[k] += static_cast<uint32_t>(__shfl_xor(static_cast<int>(c[k]), o));
    Scored s;
    const uint32_t ref_len = c[0] + c[1] + c[3];
    s.start = pos; s.end = pos + ref_len;
    const uint32_t clip = min(c[4], pos) + min(c[5], contig_len > s.end ? contig_len - s.end : 0u);
    const uint32_t common = c[1] + c[2] + clip;
    s.edit = common + c[3];
    s.ln_prob = L.lp[0] * static_cast<double>(c[0]) + L.lp[1] * static_cast<double>(c[1]) + L.lp[2] * static_cast<double>(c[2])
              + L.lp[3] * static_cast<double>(c[3]) + L.lp[4] * static_cast<double>(clip);
    s.bad = __any(bad);
    return s;
}

__device__ __forceinline__ uint64_t pos_key(uint32_t read_end, uint32_t contig, uint32_t pos) {       // encode, locs.rs:181-184
    return (static_cast<uint64_t>(read_end + 1) << 48) | (static_cast<uint64_t>(contig) << 32) | static_cast<uint64_t>(pos >> 7);
}
__device__ inline uint32_t hfind(const PairScratch& P, uint64_t key, bool* found) {
    uint32_t h = static_cast<uint32_t>(mix64(key)) & (P.hcap - 1);
    while (P.hkey[h] != 0) {
        if (P.hkey[h] == key) { *found = true; return h; }
        h = (h + 1) & (P.hcap - 1);
    }
    *found = false;
    return h;
}

// PrelimAlignments state of the pair (single lane)
struct Prelim {
    uint32_t n_alns, passable[2], best_edit[2];
    double best_lik[2];
};
constexpr uint32_t NOT_SAVED = 0xFFFFFFFFu;

// PrelimAlignments::push (locs.rs:298-344)
__device__ inline bool prelim_push(const PairScratch& P, Prelim& S, const PAln& a, uint32_t* err) {
    const uint32_t e = (a.contig_end >> 16) & 1u, contig = a.contig_end & 0xFFFFu;
    S.best_edit[e] = min(S.best_edit[e], a.edit);
    S.best_lik[e] = fmax(S.best_lik[e], a.ln_prob);
    const uint32_t new_ix = S.n_alns;
    const bool save = a.edit <= S.passable[e];
    if (new_ix == 0 && !save) return false;
    bool found;
    const uint32_t h = hfind(P, pos_key(e, contig, a.start), &found);
    auto append = [&]() {
        if (S.n_alns < P.cap_alns) P.alns[S.n_alns++] = a; else atomicMax(err, static_cast<uint32_t>(LCTY_ERR_RUNTIME));
    };
    if (found) {
        if (save) {
            if (P.hval[h].x == NOT_SAVED) { P.hval[h] = make_uint2(new_ix, a.start); append(); }
            else if (a.ln_prob > P.alns[P.hval[h].x].ln_prob) { P.alns[P.hval[h].x] = a; P.hval[h].y = a.start; }
        }
    } else {
        P.hkey[h] = pos_key(e, contig, a.start);
        if (save) { P.hval[h] = make_uint2(new_ix, a.start); append(); }
        else P.hval[h] = make_uint2(NOT_SAVED, a.start);
    }
    return save;
}

// PosCollection::get (locs.rs:245-262), the neighbour test as written there
__device__ inline bool pos_get(const PairScratch& P, uint32_t read_end, uint32_t contig, uint32_t pos, uint32_t* index) {
    const uint64_t key = pos_key(read_end, contig, pos);
    bool found;
    uint32_t h = hfind(P, key, &found);
    if (found) { *index = P.hval[h].x; return true; }
    const uint64_t key2 = (pos & 64u) == 0 ? key - 1 : key + 1;
    h = hfind(P, key2, &found);
    if (found) {
        const uint32_t sp = P.hval[h].y;
        const uint32_t d = sp > pos ? sp - pos : pos - sp;
        if ((d >> 6) != 0) { *index = P.hval[h].x; return true; }
    }
    return false;
}

struct TransferArgs {
    uint32_t cap_alns, hcap, cap_new, cap_words;
    Limits lim; uint32_t last_level;                    // what a lane can hold at this level; 1: nothing behind it
    uint32_t walk_budget;                               // phases a lane walks before the wavefront looks who needs the aligner (xfer::walk_step)
    const uint64_t* pair_list; uint64_t n_list;         // the pairs of this launch (NULL: all of the batch)
    uint64_t* redo_list; unsigned long long* redo_n;    // pairs handed to the next level
    uint8_t* scratch; size_t scratch_stride;
    // outputs
    uint32_t* new_cnt;             // [R][2] transferred alignments per read end
    uint32_t* new_words;           // [R] CIGAR words of the pair's transferred alignments
    unsigned long long* rec_cursor; unsigned long long* word_cursor;
    lcty_aln_rec* out_recs; uint64_t out_recs_cap;      // arena: per pair contiguous, end 0 then end 1, push order inside
    uint32_t* out_words; uint64_t out_words_cap;
    uint64_t* out_rec_at; uint64_t* out_word_at;        // [R]
    uint32_t* flag;                // 1 = arena overflow (retry larger), LCTY_ERR_* >= 2 otherwise
    unsigned long long* dp_cells;  // cells of the aligner's matrices, all lanes
    unsigned long long* phases;    // developer build (knob transfer_phases): shader-clock ticks of the kernel's parts, summed over the wavefronts; else NULL
    double min_weight;
    uint32_t wave_scores;          // 1: the batch has records whose CIGARs are long enough to be counted by the whole wavefront (WAVE_SCORE_FROM words)
    uint32_t dry_run;              // 1: only look where a transfer WOULD start (no similar position on the target yet): pairs with any go to
                                   // redo_list, their number of such (source, target) combinations is added to rec_cursor; nothing is walked
};

// WPS wavefronts per SIMD: 3 (168 VGPRs) for batches with long CIGARs, whose lanes live in the walk and the in-register aligner (they spill
// at 128); 4 (128 VGPRs) for short reads, whose transfers are a handful of items each and gain from the fourth wavefront
template <int WPS>
__global__ __launch_bounds__(64, WPS) void transfer_kernel(const LocusView L, const ReadsView R, const HapView H, const TransferArgs T) {
    const uint32_t lane = threadIdx.x;
    uint8_t* base = T.scratch + static_cast<size_t>(blockIdx.x) * T.scratch_stride;
    const PairScratch P = carve(base, T.cap_alns, T.hcap, T.cap_new, T.cap_words);
    Scratch LS = scratch_at(P.lanes, lane, T.lim);
    const bool paired = L.is_paired != 0;
    __shared__ Prelim S;
    __shared__ uint32_t sh_n_new, sh_words, sh_fails, sh_stop, sh_redo;
    __shared__ uint32_t lds_a[CIGAR_CHUNK * 64], lds_b[CIGAR_CHUNK * 64];   // two slices of 16 words per lane: the chunk of its CIGAR under construction, and
                                                                             // the finished CIGAR that is read meanwhile (whole, or the window read ahead): DCigar
    constexpr uint32_t SRC_WORDS = WPS == 3 ? SRC_LDS_WORDS : 64u;   // (short reads: a few words — and 16 wavefronts per CU must fit the LDS)
    __shared__ uint32_t lds_src[SRC_WORDS];                        // the source alignment's CIGAR: every lane of every chunk walks it
    // developer build: where a wavefront's time goes (0 set-up + wave scoring, 1 PrelimAlignments, 2 estimate / probe / offset / walk_init,
    // 3 walk, 4 aligner for clipped ends, 5 assemble, 6 optimize, 7 scoring of the result, 8 push, 9 hand-over)
    [[maybe_unused]] unsigned long long tph[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    auto mark = [&](int k) {
        if constexpr (kDiag) {
            if (T.phases) {
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long now = __builtin_amdgcn_s_memtime();
                __builtin_amdgcn_sched_barrier(0);
                tph[k] += now - tlast; tlast = now;
            }
        }
    };
    if constexpr (kDiag) tlast = __builtin_amdgcn_s_memtime();

    for (uint64_t ii = blockIdx.x; ii < T.n_list; ii += gridDim.x) {
        const uint64_t p = T.pair_list ? T.pair_list[ii] : ii;
        if (lane == 0) { T.new_cnt[2 * p] = T.new_cnt[2 * p + 1] = 0; T.new_words[p] = 0; }
        if (!(R.recover_w[p] >= T.min_weight)) continue;                    // locs.rs:1257 (and not a candidate at all: -1)
        const uint64_t a0 = R.aln_off[p];
        const uint32_t* cig = R.cigar + R.cigar_off[p];
        const uint2 meta = R.pair_meta[p];
        const uint32_t j2 = meta.x, n_eff = meta.y, split = min(j2, n_eff);
        const lcty_aln_rec* recs = R.recs + a0;
        const uint32_t len0 = R.mate_len[2 * p], len1 = paired ? R.mate_len[2 * p + 1] : 0u;
        for (uint32_t i = lane; i < P.hcap; i += 64) P.hkey[i] = 0;
        for (uint32_t i = lane; i < P.cap_alns; i += 64) P.seen[i] = 0;
        __syncthreads();

        // the records with long CIGARs are scored by the whole wavefront first (the lane that builds the PrelimAlignments below used to
        // read every word of every record itself: 770 k words for a 10-kb read with a record on each of 256 alleles)
        if (T.wave_scores) {
            for (uint32_t idx = 0; idx < n_eff; idx++) {
                const lcty_aln_rec rc = recs[idx];
                if (rc.n_cigar < WAVE_SCORE_FROM) continue;
                const bool primary = idx == 0 || (paired && idx == j2);
                const Scored sc = score_words_wave(L, cig + rc.cigar_rel, rc.n_cigar, rc.pos, L.allele_len[rc.contig], primary, lane);
                if (lane == 0) P.pre[idx] = make_uint4(static_cast<uint32_t>(__double2loint(sc.ln_prob)), static_cast<uint32_t>(__double2hiint(sc.ln_prob)), sc.end,
                                                       sc.edit | (sc.bad ? 0x80000000u : 0u));
            }
            __syncthreads();
        }
        mark(0);
        // ---- PrelimAlignments of the pair, as read_next_alns builds them (locs.rs:502-567), one lane: the order matters ----
        if (lane == 0) {
            S.n_alns = 0;
            for (int e = 0; e < 2; e++) { S.passable[e] = NONE32T; S.best_edit[e] = NONE32T; S.best_lik[e] = -INFINITY; }
            sh_n_new = 0; sh_words = 0; sh_redo = 0;
            for (uint32_t e = 0; e < (paired ? 2u : 1u); e++) {
                const uint32_t first = e ? j2 : 0u, last = e ? n_eff : split;
                const uint32_t read_len = e ? len1 : len0;
                for (uint32_t idx = first; idx < last; idx++) {
                    const lcty_aln_rec rc = recs[idx];
                    const bool primary = idx == first;
                    if (rc.n_cigar == 0) continue;                           // skipped with a warning, locs.rs:550-554
                    Scored sc;
                    if (T.wave_scores && rc.n_cigar >= WAVE_SCORE_FROM) {
                        const uint4 pv = P.pre[idx];
                        sc.ln_prob = __hiloint2double(static_cast<int>(pv.y), static_cast<int>(pv.x)); sc.start = rc.pos; sc.end = pv.z;
                        sc.edit = pv.w & 0x7FFFFFFFu; sc.bad = (pv.w >> 31) != 0;
                    } else sc = score_words(L, cig + rc.cigar_rel, rc.n_cigar, rc.pos, L.allele_len[rc.contig], primary);
                    if (primary) {                                           // thresholds, locs.rs:529-537
                        const uint2 gp = L.edit_lut[min(read_len, L.edit_lut_size - 1)];
                        uint32_t good = gp.x, passable = gp.y, thr = good;
                        double compl_v = 1.0;
                        if (L.short_reads) {
                            const uint32_t mid = (sc.start + sc.end) / 2;
                            const uint32_t o = L.ci_off[rc.contig], npos = L.ci_off[rc.contig + 1] - o;
                            const uint32_t i = min(mid > L.half_neighb ? mid - L.half_neighb : 0u, npos - 1);
                            compl_v = static_cast<double>(L.compl_cnt[o + i]) * L.compl_mult;
                        }
                        if (compl_v <= L.poor_compl) {
                            thr = max(good, static_cast<uint32_t>(L.poor_compl_edit * static_cast<double>(read_len)));
                            passable += thr - good;
                        }
                        S.passable[e] = passable;
                    }
                    PAln a;
                    a.ln_prob = sc.ln_prob; a.start = sc.start; a.edit = sc.edit; a.src = idx;
                    a.contig_end = rc.contig | (e << 16) | ((rc.flags & LCTY_FLAG_REVERSE) ? (1u << 17) : 0u);
                    prelim_push(P, S, a, T.flag);
                }
            }
        }
        __syncthreads();
        mark(1);
        const uint32_t n0 = S.n_alns;

        // ---- HapAlns::transfer_alignments (transfer.rs:70-140): sources in order, targets of a source across lanes ----
        for (uint32_t i = 0; i < n0 && !sh_redo; i++) {
            if (P.seen[i]) continue;                                         // uniform: written before the barrier below
            __syncthreads();
            if (lane == 0) { P.seen[i] = 1; sh_fails = 0; sh_stop = 0; }
            __syncthreads();
            const PAln sa = P.alns[i];
            const uint32_t s_contig = sa.contig_end & 0xFFFFu, e = (sa.contig_end >> 16) & 1u;
            const bool s_rev = (sa.contig_end >> 17) & 1u;
            SrcCigar src;
            if (sa.src & 0x80000000u) { const NewAln na = P.news[sa.src & 0x7FFFFFFFu]; src.raw = P.words + na.cigar_at; src.n = na.n_cigar; src.hard_to_soft = false; }
            else { src.raw = cig + recs[sa.src].cigar_rel; src.n = recs[sa.src].n_cigar; src.hard_to_soft = true; }
            // its first SRC_LDS_WORDS words go to LDS (a 10-kb ONT read: ~900): the lanes of all chunks walk this one CIGAR, each at its own item
            src.lds = lds_src; src.lds_n = T.dry_run ? 0u : min(src.n, SRC_WORDS);
            for (uint32_t k = lane; k < src.lds_n; k += 64) lds_src[k] = src.raw[k];
            __syncthreads();
            // the read in the orientation of the source alignment: the stored bases are the primary record's (MateData::new)
            const uint32_t prim = e ? j2 : 0u;
            const bool prim_rev = (recs[prim].flags & LCTY_FLAG_REVERSE) != 0;
            const uint64_t moff = R.mate_off[2 * p + e];
            Seqs Q;
            Q.w64 = reinterpret_cast<const uint64_t*>(R.bases2) + (moff >> 5); Q.nm = R.nmask + (moff >> 5);
            Q.read_len = e ? len1 : len0; Q.flip = s_rev != prim_rev;
            const uint32_t passable = S.passable[e];
            const uint32_t nb = H.best_off[s_contig + 1] - H.best_off[s_contig];
            for (uint32_t t0 = 0; t0 < nb && !sh_stop; t0 += 64) {
                const uint32_t t = t0 + lane;
                // 0 nothing, 1 similar position exists (index in `hit`), 2 failed transfer, 3 new alignment
                uint32_t kind = 0, hit = NONE32T, target = 0;
                PAln na; na.ln_prob = 0.0; na.start = 0; na.contig_end = 0; na.edit = 0; na.src = 0;
                uint32_t* const slice_a = lds_a + lane; uint32_t* const slice_b = lds_b + lane;
                DCigar out; out.init(LS.cig_a, T.lim.cigar_cap, slice_a);
                bool beyond = false;                                         // this transfer needs a lane with more scratch
                bool transferring = false, walking = false;
                Walk walk; walk.start_k = 0; walk.phase = PH_DONE;
                if (t < nb) {
                    // one record per (source contig, target): what the walk needs of the pair's alignment, in the order the targets are tried
                    const uint4 bm = H.best_meta[H.best_off[s_contig] + t];
                    target = bm.x;
                    const int dir = s_contig < target ? 0 : 1;
                    const uint2* sp = H.sparse + bm.w;
                    // find_approx_position (cigar.rs:1128-1140)
                    const uint32_t si = sa.start >> 8;
                    const uint2 s1 = sp[si], s2 = sp[si + 1];
                    const uint32_t approx = s1.y + (((sa.start & 255u) * (s2.y - s1.y)) >> 8);
                    if (pos_get(P, e, target, approx, &hit)) kind = 1;
                    else if (T.dry_run) kind = 4;
                    else {
                        // find_cigar_offset (cigar.rs:1143-1162)
                        const uint2* pos = H.positions + bm.y;
                        uint32_t ci;
                        if (s1.x == s2.x) ci = s1.x;
                        else {
                            uint32_t l = s1.x, h = s2.x + 1;
                            while (l < h) {
                                const uint32_t mid = l + (h - l) / 2;
                                const uint32_t qp = dir ? pos[mid].y : pos[mid].x;
                                if (qp <= sa.start) l = mid + 1; else h = mid;
                            }
                            ci = l - 1;
                        }
                        const uint32_t qpos_at = dir ? pos[ci].y : pos[ci].x, rpos_at = dir ? pos[ci].x : pos[ci].y;
                        Q.target = H.seqs + H.seq_off[target];
                        Q.target_len = static_cast<uint32_t>(H.seq_off[target + 1] - H.seq_off[target]);
                        LS.big = 0;
                        transferring = true;
                        walking = walk_init(walk, H.items + bm.y, bm.z, dir, sa.start, ci, qpos_at, rpos_at,
                                            src, out);
                    }
                }
                mark(2);
                if (T.dry_run) {
                    // an upper bound: the real run also finds the alignments it has transferred itself, and stops at its failures
                    const unsigned long long would = __ballot(kind == 4);
                    if (kind == 1 && hit < n0) P.seen[hit] = 1;
                    __syncthreads();
                    if (lane == 0) sh_n_new += static_cast<uint32_t>(__popcll(would));
                    __syncthreads();
                    continue;
                }
                // the walks of all lanes, resumed until none of them waits for the aligner any more: one converged call site
                {
                    Job job;
                    LS.cig_free = LS.cig_b;
                    do {
                        const uint32_t st = walking ? walk_step(walk, src, Q, out, LS, job, T.walk_budget) : WALK_DONE;
                        mark(3);
                        if (st == WALK_JOB) aligner_align<false>(Q, job.i1, job.n, job.j1, job.m, job.semiglobal, job.left_clipping, out, LS);   // the clipped ends
                        mark(4);
                        if (__any(st == WALK_ASSEMBLE)) {
                            // The stretches between anchors the walks left behind, resolved for all lanes together: a lane copies its items
                            // up to the next marker (a few items), then the lanes that stand at a marker call the aligner at one converged
                            // site. The first item behind a marker was pushed with push_checked (the anchor that follows a stretch): it is
                            // pushed that way again, behind the aligner's operations; every other item keeps its boundaries.
                            DCigar fin; fin.init(LS.cig_b, T.lim.cigar_cap, out.slice == slice_a ? slice_b : slice_a);
                            bool assembling = st == WALK_ASSEMBLE;
                            uint32_t ai = 0;
                            bool after_mark = false;
                            if (assembling) out.finish();                         // `fin` collects in the lane's other slice
                            ItemWindow win; win.init(out);
                            do {
                                // a round: every lane that is copying reads its next ITEM_WIN items (all lanes' loads in flight together), copies
                                // them up to a marker, and the lanes that stand at one call the aligner — the lanes drift apart instead of
                                // waiting for each other at every marker (waiting there: 82.6 ms for the launch instead of 74.4)
                                bool need = false;
                                uint4 jb = make_uint4(0, 0, 0, 0);
                                if (assembling) {
                                    while (win.has(ai)) {
                                        const uint2 it = win.get(ai++);
                                        if (it.x == JOB_MARK) { jb = LS.jobs[static_cast<size_t>(it.y) * LANE_STRIDE]; need = true; break; }
                                        if (after_mark) { fin.push_checked(it.x, it.y); after_mark = false; }
                                        else fin.push_raw(it);
                                    }
                                    if (!need && ai >= out.n) assembling = false;
                                }
                                if (assembling && !need) win.fill(out, ai);     // at the end of its window (the first round: of the empty one)
                                [[maybe_unused]] unsigned long long t0 = 0;
                                if constexpr (kDiag) { if (T.phases) { __builtin_amdgcn_sched_barrier(0); t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } }
                                if (need) { aligner_align<WPS == 3>(Q, jb.x, jb.y, jb.z, jb.w, 0, false, fin, LS); after_mark = true; }
                                if constexpr (kDiag) {
                                    if (T.phases) {
                                        __builtin_amdgcn_sched_barrier(0);
                                        tph[10] += __builtin_amdgcn_s_memtime() - t0; tph[11]++;
                                        tph[16] += static_cast<unsigned long long>(__popcll(__ballot(need)));
                                        uint32_t mx = need ? (jb.y + 1) * (jb.w + 1) : 0u;
                                        for (int o = 32; o > 0; o >>= 1) mx = max(mx, static_cast<uint32_t>(__shfl_xor(static_cast<int>(mx), o)));
                                        tph[17] += mx;
                                        tph[18] += __any(need && (jb.y > xfer::DP_SMALL || jb.w > xfer::DP_SMALL)) ? 1u : 0u;
                                    }
                                }
                            } while (__any(assembling));
                            if (st == WALK_ASSEMBLE) {
                                fin.rlen = out.rlen; fin.qlen = out.qlen; fin.overflow |= out.overflow;
                                out = fin;                                        // the CIGAR lives in cig_b now; cig_a is the free one
                                LS.cig_free = LS.cig_a;
                                walk.n_jobs = 0;
                            }
                            mark(5);
                        }
                        if (__any(st == WALK_OPTIMIZE)) {
                            optimize_and_finish<WPS == 3>(st == WALK_OPTIMIZE, out, Q, LS, slice_a, slice_b, kDiag && T.phases ? tph + 12 : nullptr);
                            if (st == WALK_OPTIMIZE) walk.phase = PH_DONE;
                            mark(6);
                        }
                        walking = st != WALK_DONE && st != WALK_OPTIMIZE;
                    } while (__any(walking));
                }
                out.finish();
                if (transferring) {
                    const uint32_t new_start = walk.start_k;
                    const uint32_t diff = out.rlen > out.qlen ? out.rlen - out.qlen : out.qlen - out.rlen;
                    // the lengths of an end-to-end stretch do not depend on how it is aligned: a stand-in decides the length test
                    if (out.overflow || (LS.big & 2u)) beyond = true;
                    else if (diff > passable || out.rlen < 50) kind = 2;    // MIN_ALN_SIZE
                    else if (LS.big & 1u) beyond = true;
                    else {
                        kind = 3;
                        // Alignment::new + the scoring of push(): the CIGAR goes through the same counting as a record's
                        // (a CIGAR of one chunk is still in its LDS slice: DCigar::keep)
                        const uint32_t* kept = out.keep;
                        const Scored sc = kept ? score_cigar(L, [kept](uint32_t i) { return kept[i * 64]; }, out.n, new_start, Q.target_len, false)
                                               : score_cigar(L, [&out](uint32_t i) { return *out.at(i); }, out.n, new_start, Q.target_len, false);
                        na.ln_prob = sc.ln_prob; na.start = new_start; na.edit = sc.edit;
                        na.contig_end = target | (e << 16) | (s_rev ? (1u << 17) : 0u);
                    }
                }
                mark(7);
                // the loop of transfer.rs:90-136 stops at the (transfer_fails + 1)-th failure: everything behind it did not happen
                const unsigned long long fail_mask = __ballot(kind == 2);
                const uint32_t fails_before = sh_fails + static_cast<uint32_t>(__popcll(fail_mask & ((1ull << lane) - 1ull)));
                const bool alive = t < nb && fails_before <= H.transfer_fails;
                // the first transfer that is beyond this level decides (what is behind it may depend on its outcome): if the loop gets
                // that far, the whole pair goes to the next level
                const unsigned long long beyond_mask = __ballot(beyond);
                if (beyond_mask) {
                    const int first = __ffsll(static_cast<long long>(beyond_mask)) - 1;
                    if (__shfl(alive ? 1 : 0, first)) {
                        if (lane == 0) {
                            sh_redo = 1; sh_stop = 1;
                            if (T.last_level) atomicMax(T.flag, static_cast<uint32_t>(LCTY_ERR_UNSUPPORTED));
                            else T.redo_list[atomicAdd(T.redo_n, 1ull)] = p;
                        }
                        __syncthreads();
                        break;
                    }
                }
                if (alive && kind == 1 && hit < n0) P.seen[hit] = 1;
                const bool fresh = alive && kind == 3;
                const unsigned long long fresh_mask = __ballot(fresh);
                // slots and CIGAR space in lane order
                uint32_t my_words = fresh ? out.n : 0u, word_at = my_words;
                for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(word_at, o); if (lane >= static_cast<uint32_t>(o)) word_at += v; }
                const uint32_t words_total = __shfl(word_at, 63);
                word_at = sh_words + word_at - my_words;
                const uint32_t slot = sh_n_new + static_cast<uint32_t>(__popcll(fresh_mask & ((1ull << lane) - 1ull)));
                const uint32_t n_fresh = static_cast<uint32_t>(__popcll(fresh_mask));
                const bool room = sh_n_new + n_fresh <= P.cap_new && sh_words + words_total <= P.cap_words;
                if (!room) atomicMax(T.flag, 1u);
                if (fresh && room) {
                    NewAln nn;
                    nn.ln_prob = na.ln_prob; nn.start = na.start; nn.contig_end = na.contig_end; nn.edit = na.edit;
                    nn.n_cigar = out.n; nn.cigar_at = word_at; nn.pushed = 0;
                    P.news[slot] = nn;
                    if (out.keep) for (uint32_t k = 0; k < out.n; k++) P.words[word_at + k] = out.keep[k * 64];
                    else for (uint32_t k = 0; k < out.n; k += 8) {              // eight loads in flight (one by one each waited for the store before it)
                        uint32_t v[8];
#pragma unroll
                        for (uint32_t u = 0; u < 8; u++) v[u] = k + u < out.n ? *out.at(k + u) : 0u;
#pragma unroll
                        for (uint32_t u = 0; u < 8; u++) if (k + u < out.n) P.words[word_at + k + u] = v[u];
                    }
                }
                __syncthreads();
                // PrelimAlignments::push of the chunk's new alignments (locs.rs:298-344), all lanes at once: the targets of one source are
                // distinct contigs, so their position keys are distinct and every lane's table entry is its own; what the serial loop
                // makes order-dependent is only the index an appended alignment gets, which is its rank among the appending lanes
                if (room && n_fresh) {
                    const uint32_t n_before = S.n_alns;                      // > 0: the pair has a saved primary (it would not be here otherwise)
                    uint32_t be = fresh ? na.edit : NONE32T;
                    double bl = fresh ? na.ln_prob : -INFINITY;
                    for (int o = 32; o > 0; o >>= 1) { be = min(be, static_cast<uint32_t>(__shfl_xor(static_cast<int>(be), o))); bl = fmax(bl, __shfl_xor(bl, o)); }
                    const bool save = fresh && na.edit <= S.passable[e];
                    uint32_t h = 0, existing = NOT_SAVED;
                    bool found = false;
                    const uint64_t key = pos_key(e, target, na.start);
                    if (fresh) {
                        h = static_cast<uint32_t>(mix64(key)) & (P.hcap - 1);
                        for (;;) {
                            const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&P.hkey[h]), 0ull, static_cast<unsigned long long>(key));
                            if (old == 0ull) break;
                            if (old == key) { found = true; existing = P.hval[h].x; break; }
                            h = (h + 1) & (P.hcap - 1);
                        }
                    }
                    const bool appends = save && (!found || existing == NOT_SAVED);
                    const unsigned long long app_mask = __ballot(appends);
                    const uint32_t new_ix = n_before + static_cast<uint32_t>(__popcll(app_mask & ((1ull << lane) - 1ull)));
                    const uint32_t n_app = static_cast<uint32_t>(__popcll(app_mask));
                    const bool fits = n_before + n_app <= P.cap_alns;
                    if (!fits) atomicMax(T.flag, static_cast<uint32_t>(LCTY_ERR_RUNTIME));
                    if (fresh && fits) {
                        PAln a;
                        a.ln_prob = na.ln_prob; a.start = na.start; a.contig_end = na.contig_end; a.edit = na.edit;
                        a.src = 0x80000000u | slot;
                        if (appends) { P.hval[h] = make_uint2(new_ix, a.start); P.alns[new_ix] = a; }
                        else if (save) { if (a.ln_prob > P.alns[existing].ln_prob) { P.alns[existing] = a; P.hval[h].y = a.start; } }
                        else if (!found) P.hval[h] = make_uint2(NOT_SAVED, a.start);
                        P.news[slot].pushed = 1;
                    }
                    if (lane == 0) {
                        S.best_edit[e] = min(S.best_edit[e], be); S.best_lik[e] = fmax(S.best_lik[e], bl);
                        if (fits) S.n_alns = n_before + n_app;
                    }
                }
                __syncthreads();
                if (lane == 0) {
                    if (room) { sh_n_new += n_fresh; sh_words += words_total; }
                    sh_fails += static_cast<uint32_t>(__popcll(fail_mask));
                    if (sh_fails > H.transfer_fails || !room) sh_stop = 1;
                }
                __syncthreads();
                mark(8);
            }
        }
        __syncthreads();

        if (T.dry_run) {
            if (lane == 0 && sh_n_new) {
                T.redo_list[atomicAdd(T.redo_n, 1ull)] = p;
                atomicAdd(T.rec_cursor, static_cast<unsigned long long>(sh_n_new));
            }
            __syncthreads();
            continue;
        }
        // ---- hand the transferred alignments over: per pair a contiguous run, read end 0 first, push order inside an end ----
        const uint32_t n_new = sh_redo ? 0u : sh_n_new, n_words = sh_words;
        if (n_new) {
            __shared__ unsigned long long rec_at, word_at0;
            if (lane == 0) {
                rec_at = atomicAdd(T.rec_cursor, static_cast<unsigned long long>(n_new));
                word_at0 = atomicAdd(T.word_cursor, static_cast<unsigned long long>(n_words));
            }
            __syncthreads();
            const bool room = rec_at + n_new <= T.out_recs_cap && word_at0 + n_words <= T.out_words_cap;
            if (!room) { if (lane == 0) atomicMax(T.flag, 1u); }
            else {
                // sources are taken in push order, read end 0 before read end 1, so the transferred alignments already lie end 0 first
                uint32_t c0 = 0;
                for (uint32_t k = lane; k < n_new; k += 64) {
                    const NewAln nn = P.news[k];
                    const uint32_t e2 = (nn.contig_end >> 16) & 1u;
                    c0 += e2 == 0;
                    lcty_aln_rec r;
                    r.pos = nn.start; r.contig = static_cast<uint16_t>(nn.contig_end & 0xFFFFu);
                    r.flags = static_cast<uint16_t>(LCTY_FLAG_SECONDARY | (((nn.contig_end >> 17) & 1u) ? LCTY_FLAG_REVERSE : 0) | (e2 ? LCTY_FLAG_MATE2 : 0));
                    r.n_cigar = nn.n_cigar; r.cigar_rel = nn.cigar_at;         // relative to the pair's block of new words
                    T.out_recs[rec_at + k] = r;
                }
                for (int o = 32; o > 0; o >>= 1) c0 += __shfl_xor(static_cast<int>(c0), o);
                if (lane == 0) {
                    T.new_cnt[2 * p] = c0; T.new_cnt[2 * p + 1] = n_new - c0; T.new_words[p] = n_words;
                    T.out_rec_at[p] = rec_at; T.out_word_at[p] = word_at0;
                }
                for (uint32_t k = lane; k < n_words; k += 64) T.out_words[word_at0 + k] = P.words[k];
            }
        }
        __syncthreads();
        mark(9);
    }
    if constexpr (kDiag) {
        if (T.phases && lane == 0) for (int k = 0; k < 20; k++) atomicAdd(&T.phases[k], tph[k]);
    }
    // aligner work of this wavefront (launches that are repeated with larger arenas count again: it is what the device did)
    unsigned long long cells = LS.cells;
    for (int o = 32; o > 0; o >>= 1) cells += __shfl_xor(cells, o);
    if (lane == 0 && cells) atomicAdd(T.dp_cells, cells);
}

// merged record table: per pair [end-0 originals][end-0 transferred][end-1 originals][end-1 transferred][anything behind n_eff]
__global__ __launch_bounds__(64) void merge_kernel(const ReadsView R, const uint32_t* __restrict__ new_cnt, const uint32_t* __restrict__ new_words,
                                                   const uint64_t* __restrict__ rec_at, const uint64_t* __restrict__ word_at,
                                                   const lcty_aln_rec* __restrict__ xrecs, const uint32_t* __restrict__ xwords,
                                                   const uint64_t* __restrict__ m_aln_off, const uint64_t* __restrict__ m_cigar_off,
                                                   lcty_aln_rec* __restrict__ m_recs, uint32_t* __restrict__ m_cigar, uint2* __restrict__ m_meta) {
    const uint32_t lane = threadIdx.x;
    for (uint64_t p = blockIdx.x; p < R.n_pairs; p += gridDim.x) {
        const uint64_t a0 = R.aln_off[p], c0 = R.cigar_off[p];
        const uint32_t n_old = static_cast<uint32_t>(R.aln_off[p + 1] - a0), w_old = static_cast<uint32_t>(R.cigar_off[p + 1] - c0);
        const uint2 meta = R.pair_meta[p];
        const uint32_t j2 = meta.x, n_eff = meta.y, split = min(j2, n_eff);
        const uint32_t n0 = new_cnt[2 * p], n1 = new_cnt[2 * p + 1];
        lcty_aln_rec* out = m_recs + m_aln_off[p];
        uint32_t* cw = m_cigar + m_cigar_off[p];
        for (uint32_t k = lane; k < w_old; k += 64) cw[k] = R.cigar[c0 + k];
        for (uint32_t k = lane; k < new_words[p]; k += 64) cw[w_old + k] = xwords[word_at[p] + k];
        for (uint32_t k = lane; k < n_old + n0 + n1; k += 64) {
            lcty_aln_rec r;
            if (k < split) r = R.recs[a0 + k];
            else if (k < split + n0) { r = xrecs[rec_at[p] + (k - split)]; r.cigar_rel += w_old; }
            else if (k < n_eff + n0) r = R.recs[a0 + (k - n0)];
            else if (k < n_eff + n0 + n1) { r = xrecs[rec_at[p] + n0 + (k - n_eff - n0)]; r.cigar_rel += w_old; }
            else r = R.recs[a0 + (k - n0 - n1)];
            out[k] = r;
        }
        if (lane == 0) m_meta[p] = make_uint2(j2 + n0, n_eff + n0 + n1);
    }
}

}  // namespace lcty

using namespace lcty;

extern "C" {

// HapAlns::new / add / sort_best_ixs (transfer.rs:33-67) + CigarIndex::new (cigar.rs:1099-1126) on the host
int32_t lcty_locus_set_hap_alns(lcty_locus* loc, uint32_t n_entries, const uint32_t* id1, const uint32_t* id2, const uint64_t* cigar_off,
                                const uint32_t* cigar, const uint32_t* n_matches, const uint32_t* aln_len, uint32_t transfer_fails,
                                double max_div) {
    return guarded([&] {
        if (!loc || (n_entries && (!id1 || !id2 || !cigar_off || !cigar || !n_matches || !aln_len))) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        const uint32_t A = loc->n_alleles;
        struct Cell { std::vector<uint2> items, positions; std::vector<uint2> sparse[2]; uint32_t rlen = 0, qlen = 0; };
        std::vector<uint32_t> cell_of(static_cast<size_t>(A) * A, NONE32T);
        std::vector<Cell> cells;
        struct Best { uint32_t id, n_matches, order; };
        std::vector<std::vector<Best>> best(A);
        auto cq = [](uint32_t op) { return op == 0 || op == 7 || op == 8 || op == 1 || op == 4; };
        auto cr = [](uint32_t op) { return op == 0 || op == 7 || op == 8 || op == 2; };
        // which entries become cells (serial: the first alignment of a pair wins), then the cells themselves on the host's cores — the
        // CIGAR walk of 32 640 pairs of 50-kb haplotypes is 0.3 s on one
        std::vector<uint32_t> taken;
        for (uint32_t t = 0; t < n_entries; t++) {
            const uint32_t q = id1[t], r = id2[t];
            if (q >= A || r >= A) fail(LCTY_ERR_INVALID_INPUT, "haplotype alignment %u refers to a contig out of range", t);
            if (q == r) continue;
            const uint32_t lo = std::min(q, r), hi = std::max(q, r);
            if (cell_of[static_cast<size_t>(lo) * A + hi] != NONE32T) continue;                  // first alignment of a pair wins
            const double div = aln_len[t] == 0 ? std::numeric_limits<double>::infinity()
                                               : static_cast<double>(aln_len[t] - n_matches[t]) / static_cast<double>(aln_len[t]);     // paf.rs:201-208
            if (div > max_div) continue;
            if (cigar_off[t + 1] == cigar_off[t]) continue;
            // (an entry that turns out to be malformed below fails the call, as it did when the cells were built one by one)
            cell_of[static_cast<size_t>(lo) * A + hi] = static_cast<uint32_t>(taken.size());
            taken.push_back(t);
            best[q].push_back(Best{r, n_matches[t], static_cast<uint32_t>(best[q].size())});
            best[r].push_back(Best{q, n_matches[t], static_cast<uint32_t>(best[r].size())});
        }
        cells.resize(taken.size());
        const uint32_t n_threads = static_cast<uint32_t>(std::max<size_t>(1, std::min<size_t>({taken.size() / 64 + 1, 16, std::thread::hardware_concurrency()})));
        std::vector<uint32_t> bad_entry(n_threads, NONE32T), bad_kind(n_threads, 0);
        auto build = [&](uint32_t tid) {
            for (size_t ci = tid; ci < taken.size(); ci += n_threads) {
                const uint32_t t = taken[ci], q = id1[t], r = id2[t];
                const uint32_t lo = std::min(q, r), hi = std::max(q, r);
                Cell& c = cells[ci];
                bool ok_ops = true;
                c.items.reserve(cigar_off[t + 1] - cigar_off[t]); c.positions.reserve(cigar_off[t + 1] - cigar_off[t]);
                for (uint64_t k = cigar_off[t]; k < cigar_off[t + 1]; k++) {
                    uint32_t op = cigar[k] & 15u;
                    const uint32_t len = cigar[k] >> 4;
                    if (!(op == 7 || op == 8 || op == 1 || op == 2 || op == 0)) { ok_ops = false; break; }
                    if (q > r) op = op == 1 ? 2u : (op == 2 ? 1u : op);                         // Cigar::invert: query = lower id
                    c.items.push_back(make_uint2(op, len));
                }
                if (!ok_ops) { if (t < bad_entry[tid]) { bad_entry[tid] = t; bad_kind[tid] = 1; } continue; }
                // the alignment has to cover both haplotypes (full_positive_alignment, paf.rs:211-216)
                uint32_t qpos = 0, rpos = 0;
                for (size_t k = 0; k < c.items.size(); k++) {
                    const uint32_t op = c.items[k].x, len = c.items[k].y;
                    c.positions.push_back(make_uint2(qpos, rpos));
                    const uint32_t old_q = qpos;
                    auto upd = [&](std::vector<uint2>& v, uint32_t pos1, uint32_t pos2, bool other) {   // update_sparse_index, cigar.rs:972-986
                        const uint32_t last = (pos1 + len - 1) >> 8;
                        for (uint32_t i = static_cast<uint32_t>(v.size()); i <= last; i++)
                            v.push_back(make_uint2(static_cast<uint32_t>(k), pos2 + (other ? (i << 8) - pos1 : 0u)));
                    };
                    if (cq(op)) { upd(c.sparse[0], qpos, rpos, cr(op)); qpos += len; }
                    if (cr(op)) { upd(c.sparse[1], rpos, old_q, cq(op)); rpos += len; }
                }
                c.qlen = qpos; c.rlen = rpos;
                if (qpos != loc->allele_len[lo] || rpos != loc->allele_len[hi]) { if (t < bad_entry[tid]) { bad_entry[tid] = t; bad_kind[tid] = 2; } continue; }
                c.sparse[0].push_back(make_uint2(static_cast<uint32_t>(c.items.size() - 1), c.rlen));
                c.sparse[1].push_back(make_uint2(static_cast<uint32_t>(c.items.size() - 1), c.qlen));
            }
        };
        auto in_parallel = [&](auto&& fn) {
            // a thread that cannot be started: its share is done here (every thread that did start is joined)
            std::vector<std::thread> th;
            std::vector<uint32_t> here;
            for (uint32_t tid = 1; tid < n_threads; tid++) {
                try { th.emplace_back(fn, tid); }
                catch (const std::system_error&) { here.push_back(tid); }
            }
            fn(0u);
            for (uint32_t tid : here) fn(tid);
            for (auto& x : th) x.join();
        };
        in_parallel(build);
        {
            uint32_t first_bad = NONE32T, kind = 0;
            for (uint32_t tid = 0; tid < n_threads; tid++) if (bad_entry[tid] < first_bad) { first_bad = bad_entry[tid]; kind = bad_kind[tid]; }
            if (kind == 1) fail(LCTY_ERR_INVALID_DATA, "haplotype alignment %u: unsupported CIGAR operation", first_bad);
            if (kind == 2) {
                const uint32_t t = first_bad, lo = std::min(id1[t], id2[t]), hi = std::max(id1[t], id2[t]);
                const Cell& c = cells[cell_of[static_cast<size_t>(lo) * A + hi]];
                fail(LCTY_ERR_INVALID_DATA, "haplotype alignment %u does not cover both sequences (%u/%u vs %u/%u)", t, c.qlen, c.rlen, loc->allele_len[lo],
                     loc->allele_len[hi]);
            }
        }
        for (auto& v : best) std::stable_sort(v.begin(), v.end(), [](const Best& a, const Best& b) { return a.n_matches > b.n_matches; });
        // flatten: offsets serially, the copies on the same threads
        std::vector<uint32_t> item_off(cells.size() + 1, 0), sparse_off(2 * cells.size() + 1, 0), best_off(A + 1, 0), best_ids;
        for (size_t c = 0; c < cells.size(); c++) {
            item_off[c + 1] = item_off[c] + static_cast<uint32_t>(cells[c].items.size());
            sparse_off[2 * c + 1] = sparse_off[2 * c] + static_cast<uint32_t>(cells[c].sparse[0].size());
            sparse_off[2 * c + 2] = sparse_off[2 * c + 1] + static_cast<uint32_t>(cells[c].sparse[1].size());
        }
        std::vector<uint2> items(item_off[cells.size()]), positions(item_off[cells.size()]), sparse(sparse_off[2 * cells.size()]);
        in_parallel([&](uint32_t tid) {
            for (size_t c = tid; c < cells.size(); c += n_threads) {
                std::copy(cells[c].items.begin(), cells[c].items.end(), items.begin() + item_off[c]);
                std::copy(cells[c].positions.begin(), cells[c].positions.end(), positions.begin() + item_off[c]);
                for (int d = 0; d < 2; d++) std::copy(cells[c].sparse[d].begin(), cells[c].sparse[d].end(), sparse.begin() + sparse_off[2 * c + d]);
            }
        });
        std::vector<uint4> best_meta;
        for (uint32_t a = 0; a < A; a++) {
            for (const Best& b : best[a]) {
                best_ids.push_back(b.id);
                const uint32_t lo = std::min(a, b.id), hi = std::max(a, b.id), cell = cell_of[static_cast<size_t>(lo) * A + hi];
                best_meta.push_back(make_uint4(b.id, item_off[cell], item_off[cell + 1] - item_off[cell], sparse_off[2 * cell + (a < b.id ? 0 : 1)]));
            }
            best_off[a + 1] = static_cast<uint32_t>(best_ids.size());
        }
        lcty_ctx* ctx = loc->ctx;
        ctx->activate();
        hipStream_t s = ctx->stream;
        auto up32 = [&](DevBuf<uint32_t>& d, const std::vector<uint32_t>& v) { d.alloc(std::max<size_t>(v.size(), 1)); d.upload(v.data(), v.size(), s); };
        auto up2 = [&](DevBuf<uint2>& d, const std::vector<uint2>& v) { d.alloc(std::max<size_t>(v.size(), 1)); d.upload(v.data(), v.size(), s); };
        up32(loc->d_hap_cell_of, cell_of); up32(loc->d_hap_item_off, item_off); up32(loc->d_hap_sparse_off, sparse_off);
        up32(loc->d_hap_best_off, best_off); up32(loc->d_hap_best_ids, best_ids);
        up2(loc->d_hap_items, items); up2(loc->d_hap_positions, positions); up2(loc->d_hap_sparse, sparse);
        loc->d_hap_best_meta.alloc(std::max<size_t>(best_meta.size(), 1)); loc->d_hap_best_meta.upload(best_meta.data(), best_meta.size(), s);
        LCTY_HIP(hipStreamSynchronize(s));
        loc->hap_transfer_fails = transfer_fails; loc->hap_cells = static_cast<uint32_t>(cells.size());
        loc->has_hap_alns = true;
    });
}

// recover_and_group_alignments' transfer step (locs.rs:1255-1260) for the whole batch; lcty_score_reads before and after
int32_t lcty_recover_alignments(lcty_reads* reads, uint64_t* n_recovered) {
    return guarded([&] {
        if (!reads) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        lcty_locus* loc = reads->locus;
        lcty_ctx* ctx = reads->ctx;
        if (!loc->has_hap_alns) fail(LCTY_ERR_INVALID_INPUT, "lcty_locus_set_hap_alns has not been called");
        if (reads->counted) fail(LCTY_ERR_UNSUPPORTED, "alignment recovery walks the CIGARs of the records: not available for a batch of counted alignments");
        if (!reads->scored) fail(LCTY_ERR_INVALID_INPUT, "lcty_score_reads first: recovery looks at the read pairs the first pass lets through");
        ctx->activate();
        reads->check_device_error();
        hipStream_t s = ctx->stream;
        // the pairs whose records are on the device: the whole batch, or the current chunk of a streaming batch (reads->view()
        // addresses them from 0; recovery then sits between the two scoring passes of every chunk)
        const uint64_t R = reads->n_pairs - reads->raw_first;
        if (n_recovered) *n_recovered = 0;
        if (R == 0) return;
        const uint32_t A = loc->n_alleles;
        HapView H{};
        H.A = A; H.transfer_fails = loc->hap_transfer_fails;
        H.cell_of = loc->d_hap_cell_of.p; H.item_off = loc->d_hap_item_off.p; H.items = loc->d_hap_items.p; H.positions = loc->d_hap_positions.p;
        H.sparse_off = loc->d_hap_sparse_off.p; H.sparse = loc->d_hap_sparse.p; H.best_off = loc->d_hap_best_off.p; H.best_ids = loc->d_hap_best_ids.p; H.best_meta = loc->d_hap_best_meta.p;
        H.seqs = loc->d_seqs.p; H.seq_off = loc->d_seq_off.p;

        DevBuf<uint32_t> d_new_cnt, d_new_words, d_flag;
        DevBuf<uint64_t> d_rec_at, d_word_at, d_list_a, d_list_b;
        DevBuf<unsigned long long> d_cursors;                                      // records, words, pairs for the next level, aligner cells
        d_new_cnt.alloc(2 * R); d_new_words.alloc(R); d_rec_at.alloc(R); d_word_at.alloc(R); d_flag.alloc(1); d_cursors.alloc(4 + 20);
        LCTY_HIP(hipMemsetAsync(d_cursors.p + 3, 0, 21 * sizeof(unsigned long long), s));
        const bool phases = ctx->diag_knob("transfer_phases", 0) != 0;             // developer build only
        d_list_a.alloc(R); d_list_b.alloc(R);
        // Levels of lane scratch. Level 0 is sized for the reads of the batch (short reads: transferred CIGARs of <= 192 items,
        // stretches between anchors of <= 255 bases) and takes every pair at full occupancy; a pair with a transfer that needs more
        // is repeated at the next level, which has fewer lanes in flight.
        const uint32_t rec_cigar = std::max<uint32_t>(reads->max_cigar_per_rec, 1);
        std::vector<Limits> levels;
        {
            Limits l0; l0.cigar_cap = (std::max<uint32_t>(192, 2 * rec_cigar + 128) + 15u) & ~15u; l0.dp_dim = 255; l0.dp_cells = 32768;      // whole chunks (DCigar)
            Limits l1; l1.cigar_cap = std::max<uint32_t>(2048, 4 * l0.cigar_cap); l1.dp_dim = 2047; l1.dp_cells = 1u << 20;
            Limits l2; l2.cigar_cap = 4 * l1.cigar_cap; l2.dp_dim = 16383; l2.dp_cells = 1u << 26;
            levels = {l0, l1, l2};
            levels.resize(std::min<size_t>(levels.size(), static_cast<size_t>(std::max<int64_t>(1, ctx->knob("transfer_levels", 3)))));
        }
        // long reads need tens of MB per wavefront (two CIGARs per lane, the words of every transferred alignment of the read): the more
        // wavefronts fit, the better their gathers overlap — up to half of what is free on the device, at least 24 GB
        uint64_t scratch_budget = 24ull << 30;
        {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
                scratch_budget = std::max<uint64_t>(scratch_budget, std::min<uint64_t>((free_b + ctx->transfer_scratch.n) / 2, 128ull << 30));      // what the context already holds counts as free
        }
        if (ctx->knob("transfer_scratch_mb", 0) > 0) scratch_budget = std::max<uint64_t>(64, static_cast<uint64_t>(ctx->knob("transfer_scratch_mb", 0))) << 20;
        // one wavefront per workgroup; three per SIMD for long CIGARs (launch bounds: 168 VGPRs; the walk and its in-register aligner spill at
        // 128, and a fourth wavefront bought nothing there — 6 144 ONT reads x 256 alleles: 87.8 ms with 12 per CU, 88.7 with 16; six or eight
        // per SIMD at 80 / 64 VGPRs: 155-169 ms), four for short reads (262 144 Illumina pairs x 256 alleles: 50 ms at three, 40 at four)
        const bool long_cigars = rec_cigar > 256;                             // (the longest CIGAR of a record: 150-base reads stay below, a 10-kb ONT read has ~900 words)
        uint32_t waves = long_cigars ? 12 : 16;
        waves = static_cast<uint32_t>(std::max<int64_t>(1, ctx->knob("transfer_waves", waves)));
        const uint32_t max_blocks = static_cast<uint32_t>(ctx->props.multiProcessorCount) * waves;

        uint32_t cap_new = std::min<uint32_t>(std::max<uint32_t>(4 * A, 64), 1u << 15);
        cap_new = static_cast<uint32_t>(std::max<int64_t>(1, ctx->knob("transfer_cap_new", cap_new)));    // lcty_ctx_set_knob: tests exercise the retries
        const uint32_t words_per_new = std::max<uint32_t>(24, std::min<uint32_t>(levels[0].cigar_cap, rec_cigar + 32));
        // every record can reach every other contig once; beyond 1.5 G records (24 GB) the first launch only counts and the second one fits
        uint64_t arena_recs = std::min<uint64_t>(reads->n_recs * static_cast<uint64_t>(A > 1 ? A - 1 : 1) + 1024, 1500ull << 20);
        uint64_t arena_words = std::min<uint64_t>(arena_recs * std::max<uint32_t>(4, rec_cigar + 8), 6ull << 30);
        if (ctx->knob("transfer_arena", 0) > 0) { arena_recs = static_cast<uint64_t>(ctx->knob("transfer_arena", 0)); arena_words = 2 * arena_recs; }
        std::lock_guard<std::mutex> scratch_guard(ctx->scratch_mutex);
        DevBuf<uint8_t>& d_scratch = ctx->transfer_scratch;
        DevBuf<uint8_t>& d_xrecs_bytes = ctx->transfer_recs; DevBuf<uint32_t>& d_xwords = ctx->transfer_words;     // kept between calls, as the scratch
        lcty_aln_rec* xrecs = nullptr;
        unsigned long long cursors[3] = {0, 0, 0};
        // First a look at where transfers would start at all: the position estimate on every target and the probe of the pair's position
        // set, nothing walked (a few KB of scratch per wavefront). A batch whose mapper has already reached every allele — candidate
        // generation on the device, lcty_map_long.hip — has nothing to recover, and a call that sized its lane scratch and arenas for
        // "every record onto every other contig" paid seconds of allocation for it (1.5 s at 2 048 10-kb reads x 256 alleles). The
        // walk then takes the pairs that have something to do, with arenas for about what this pass counted.
        DevBuf<uint64_t> d_list_0;
        uint64_t n_first = 0, would = 0;
        {
            const uint32_t cap_alns = reads->max_recs_per_pair + 1;
            uint32_t hcap = 64;
            while (hcap < 2 * cap_alns + 2) hcap <<= 1;
            Limits none; none.cigar_cap = 1; none.dp_dim = 0; none.dp_cells = 0;
            const size_t stride = pair_scratch_bytes(cap_alns, hcap, 1, 1, none);
            const uint32_t blocks = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>({R, max_blocks, scratch_budget / stride})));
            if (d_scratch.n < stride * blocks) d_scratch.alloc(stride * blocks);
            d_list_0.alloc(R);
            d_flag.zero(s); LCTY_HIP(hipMemsetAsync(d_cursors.p, 0, 3 * sizeof(unsigned long long), s));
            TransferArgs T{};
            T.cap_alns = cap_alns; T.hcap = hcap; T.cap_new = 1; T.cap_words = 1; T.lim = none; T.last_level = 1; T.walk_budget = 1;
            T.pair_list = nullptr; T.n_list = R; T.redo_list = d_list_0.p; T.redo_n = d_cursors.p + 2;
            T.scratch = d_scratch.p; T.scratch_stride = stride;
            T.new_cnt = d_new_cnt.p; T.new_words = d_new_words.p; T.rec_cursor = d_cursors.p; T.word_cursor = d_cursors.p + 1;
            T.out_recs = nullptr; T.out_recs_cap = 0; T.out_words = nullptr; T.out_words_cap = 0;
            T.out_rec_at = d_rec_at.p; T.out_word_at = d_word_at.p; T.flag = d_flag.p; T.dp_cells = d_cursors.p + 3; T.min_weight = loc->prm.min_weight;
            T.phases = nullptr;
            T.dry_run = 1; T.wave_scores = reads->max_cigar_per_rec >= WAVE_SCORE_FROM ? 1u : 0u;
            ctx->timed(LCTY_K_TRANSFER, [&] {
                hipLaunchKernelGGL(transfer_kernel<4>, dim3(blocks), dim3(64), 0, s, loc->view(), reads->view(), H, T);
            });
            LCTY_HIP(hipGetLastError());
            uint32_t flag = 0;
            d_flag.download(&flag, 1, s);
            d_cursors.download(cursors, 3, s);
            LCTY_HIP(hipStreamSynchronize(s));
            if (flag > 1) fail(static_cast<int32_t>(flag), "alignment recovery failed on the device");
            would = cursors[0]; n_first = cursors[2];
            reads->recover_level_pairs[0] = reads->recover_level_pairs[1] = reads->recover_level_pairs[2] = 0;
            // flag 1: a pair's alignments or positions did not fit the small scratch of this pass, so the pass may have left pairs off its
            // list: the walk then takes every pair, with the arenas sized as before the pass existed
            const bool look_incomplete = flag == 1;
            if (look_incomplete) {
                std::vector<uint64_t> all(R);
                for (uint64_t i = 0; i < R; i++) all[i] = i;
                d_list_0.upload(all.data(), R, s);
                LCTY_HIP(hipStreamSynchronize(s));
                n_first = R; would = arena_recs;
            }
            if (n_first == 0) { reads->recover_dp_cells = 0; return; }                 // every target of every alignment is there already
            // the pairs in batch order again (the pass appended them as its wavefronts got to them): the walk's arenas then fill in the
            // order they always did
            std::vector<uint64_t> lst(n_first);
            d_list_0.download(lst.data(), n_first, s);
            LCTY_HIP(hipStreamSynchronize(s));
            std::sort(lst.begin(), lst.end());
            d_list_0.upload(lst.data(), n_first, s);
            // `would` is an estimate, not a bound: the walk stops a source at transfer_fails failures, marks fewer targets as seen than this
            // pass did and can then try sources this pass skipped. An arena that turns out too small is what the retry loop below is for.
            arena_recs = std::min<uint64_t>(arena_recs, would + 1024);
            arena_words = std::min<uint64_t>(arena_words, arena_recs * std::max<uint32_t>(4, rec_cigar + 8));
        }
        for (int attempt = 0;; attempt++) {
            if (attempt > 12) fail(LCTY_ERR_RUNTIME, "alignment recovery: arenas keep overflowing");
            const uint32_t cap_alns = reads->max_recs_per_pair + cap_new;
            uint32_t hcap = 64;
            while (hcap < 2 * cap_alns + 2) hcap <<= 1;
            const uint64_t cap_words64 = static_cast<uint64_t>(cap_new) * words_per_new;
            const uint32_t cap_words = static_cast<uint32_t>(std::min<uint64_t>(cap_words64, 1u << 30));
            d_xrecs_bytes.ensure(arena_recs * sizeof(lcty_aln_rec)); d_xwords.ensure(arena_words);
            xrecs = reinterpret_cast<lcty_aln_rec*>(d_xrecs_bytes.p);
            d_flag.zero(s); LCTY_HIP(hipMemsetAsync(d_cursors.p, 0, 3 * sizeof(unsigned long long), s));      // the aligner's cell count runs on
            uint32_t flag = 0;
            reads->recover_level_pairs[0] = reads->recover_level_pairs[1] = reads->recover_level_pairs[2] = 0;
            const uint64_t* list = d_list_0.p;
            uint64_t n_list = n_first;
            for (size_t lv = 0; lv < levels.size() && n_list; lv++) {
                const Limits lim = levels[lv];
                const size_t stride = pair_scratch_bytes(cap_alns, hcap, cap_new, cap_words, lim);
                uint32_t blocks = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>({n_list, max_blocks, scratch_budget / stride})));
                // whole rounds: a wavefront takes the pairs blockIdx, blockIdx + gridDim, ... and a pair costs tens of milliseconds here, so
                // 1.5 rounds (6 144 long reads on 4 096 wavefronts) take as long as two — the same two rounds on 3 072 wavefronts run faster each
                blocks = static_cast<uint32_t>((n_list + (n_list + blocks - 1) / blocks - 1) / ((n_list + blocks - 1) / blocks));
                if (d_scratch.n < stride * blocks) d_scratch.alloc(stride * blocks);
                TransferArgs T{};
                T.cap_alns = cap_alns; T.hcap = hcap; T.cap_new = cap_new; T.cap_words = cap_words;
                T.lim = lim; T.last_level = lv + 1 == levels.size();
                T.walk_budget = 1u << 30;                             // the walk never stops to look who needs the aligner (round 3: any budget was slower)
                T.dry_run = 0; T.wave_scores = reads->max_cigar_per_rec >= WAVE_SCORE_FROM ? 1u : 0u;
                T.pair_list = list; T.n_list = n_list;
                uint64_t* next = (lv % 2 == 0) ? d_list_a.p : d_list_b.p;
                T.redo_list = next; T.redo_n = d_cursors.p + 2;
                T.scratch = d_scratch.p; T.scratch_stride = stride;
                T.new_cnt = d_new_cnt.p; T.new_words = d_new_words.p; T.rec_cursor = d_cursors.p; T.word_cursor = d_cursors.p + 1;
                T.out_recs = xrecs; T.out_recs_cap = arena_recs; T.out_words = d_xwords.p; T.out_words_cap = arena_words;
                T.out_rec_at = d_rec_at.p; T.out_word_at = d_word_at.p; T.flag = d_flag.p; T.dp_cells = d_cursors.p + 3; T.min_weight = loc->prm.min_weight;
                T.phases = phases ? d_cursors.p + 4 : nullptr;
                LCTY_HIP(hipMemsetAsync(d_cursors.p + 2, 0, sizeof(unsigned long long), s));
                ctx->timed(LCTY_K_TRANSFER, [&] {
                    if (long_cigars) hipLaunchKernelGGL(transfer_kernel<3>, dim3(blocks), dim3(64), 0, s, loc->view(), reads->view(), H, T);
                    else hipLaunchKernelGGL(transfer_kernel<4>, dim3(blocks), dim3(64), 0, s, loc->view(), reads->view(), H, T);
                });
                LCTY_HIP(hipGetLastError());
                d_flag.download(&flag, 1, s);
                d_cursors.download(cursors, 3, s);
                LCTY_HIP(hipStreamSynchronize(s));
                if (flag != 0) break;
                list = next; n_list = cursors[2];
                reads->recover_level_pairs[std::min<size_t>(lv, 2)] = T.n_list;
            }
            if (flag == 0) break;
            if (flag == LCTY_ERR_UNSUPPORTED)
                fail(LCTY_ERR_UNSUPPORTED, "alignment recovery: a stretch between anchors beyond %u bases / %u cells, or a transferred CIGAR of more than %u "
                     "operations", levels.back().dp_dim, levels.back().dp_cells, levels.back().cigar_cap);
            if (flag != 1) fail(static_cast<int32_t>(flag), "alignment recovery failed on the device");
            cap_new = std::min<uint32_t>(cap_new * 4, 1u << 20);                 // some arena was too small: larger, again
            arena_recs = std::max<uint64_t>(arena_recs * 2, cursors[0] + 1024); arena_words = std::max<uint64_t>(arena_words * 2, cursors[1] + 1024);
        }
        const uint64_t total_new = cursors[0], total_words = cursors[1];
        {
            unsigned long long cells = 0;
            d_cursors.download(&cells, 1, s, 3);
            LCTY_HIP(hipStreamSynchronize(s));
            reads->recover_dp_cells = cells;
        }
        if (phases) {
            unsigned long long t[20];
            d_cursors.download(t, 20, s, 4);
            LCTY_HIP(hipStreamSynchronize(s));
            unsigned long long sum = 0;
            for (int k = 0; k < 10; k++) sum += t[k];
            static const char* names[10] = {"set-up + wave scoring", "PrelimAlignments", "estimate / probe / offset / walk_init", "walk", "aligner (clipped ends)",
                                            "assemble", "optimize", "scoring of the result", "push", "hand-over"};
            for (int k = 0; k < 10; k++)
                std::fprintf(stderr, "[transfer phases] %-40s %6.2f %%  %llu\n", names[k], sum ? 100.0 * static_cast<double>(t[k]) / static_cast<double>(sum) : 0.0, t[k]);
            auto rounds = [&](const char* what, unsigned long long ticks, unsigned long long n, unsigned long long lanes, unsigned long long cells) {
                const double r = static_cast<double>(std::max<unsigned long long>(n, 1));
                std::fprintf(stderr, "[transfer phases] %s: %.2f %% of all inside the aligner; %llu rounds, %.1f lanes with a stretch per round, largest matrix of a "
                             "round %.1f cells on average, %.0f ticks per round\n", what, 100.0 * static_cast<double>(ticks) / static_cast<double>(sum), n,
                             static_cast<double>(lanes) / r, static_cast<double>(cells) / r, static_cast<double>(ticks) / r);
            };
            rounds("assemble", t[10], t[11], t[16], t[17]);
            rounds("optimize", t[12], t[13], t[14], t[15]);
            std::fprintf(stderr, "[transfer phases] assemble rounds with a stretch beyond %u bases in some lane: %llu; ticks in all: %llu\n", xfer::DP_SMALL, t[18], sum);
        }
        if (n_recovered) *n_recovered = total_new;
        if (total_new == 0) return;

        // merged tables
        std::vector<uint32_t> new_cnt(2 * R), new_words(R);
        std::vector<uint64_t> aln_off(R + 1), cigar_off(R + 1);
        d_new_cnt.download(new_cnt.data(), 2 * R, s); d_new_words.download(new_words.data(), R, s);
        reads->d_aln_off.download(aln_off.data(), R + 1, s); reads->d_cigar_off.download(cigar_off.data(), R + 1, s);
        LCTY_HIP(hipStreamSynchronize(s));
        std::vector<uint64_t> m_aln(R + 1, 0), m_cig(R + 1, 0);
        uint64_t max_recs = 0, max_cig = 0;
        for (uint64_t p = 0; p < R; p++) {
            const uint64_t nr = aln_off[p + 1] - aln_off[p] + new_cnt[2 * p] + new_cnt[2 * p + 1];
            const uint64_t nw = cigar_off[p + 1] - cigar_off[p] + new_words[p];
            m_aln[p + 1] = m_aln[p] + nr; m_cig[p + 1] = m_cig[p] + nw;
            max_recs = std::max(max_recs, nr); max_cig = std::max(max_cig, nw);
        }
        if (max_recs > 65535) fail(LCTY_ERR_UNSUPPORTED, "more than 65535 alignments of one read pair after recovery");
        // the merged tables: a resident batch gets new ones, exactly as large as they are full (the old ones are freed below); a streaming
        // batch merges into its spare set and swaps (lcty_objects.hpp), both sets at least a chunk's capacity, so that neither this call
        // nor the next chunk's append allocates
        const bool keep = reads->streaming;
        DevBuf<uint64_t> l_aln, l_cig; DevBuf<lcty_aln_rec> l_recs; DevBuf<uint32_t> l_cigar; DevBuf<uint2> l_meta;
        DevBuf<uint64_t>& d_m_aln = keep ? reads->spare_aln_off : l_aln; DevBuf<uint64_t>& d_m_cig = keep ? reads->spare_cigar_off : l_cig;
        DevBuf<lcty_aln_rec>& d_m_recs = keep ? reads->spare_recs : l_recs; DevBuf<uint32_t>& d_m_cigar = keep ? reads->spare_cigar : l_cigar;
        DevBuf<uint2>& d_m_meta = keep ? reads->spare_pair_meta : l_meta;
        const uint64_t raw_pairs_cap = std::max<uint64_t>(reads->streaming ? reads->cap_raw_pairs : reads->cap_pairs, R);
        if (keep) {
            d_m_aln.ensure(raw_pairs_cap + 1); d_m_cig.ensure(raw_pairs_cap + 1); d_m_meta.ensure(raw_pairs_cap);
            // (grow-only, by a sixteenth at a time: the merged chunks of a batch differ by a per cent or two, and without head-room every
            // chunk a little larger than all before it was a fresh 22-GB allocation — eight of them in 31 chunks, a third of a second each)
            auto grown = [](uint64_t need) { return need + need / 16; };
            const uint64_t need_recs = std::max<uint64_t>(m_aln[R] + 1, reads->chunk_cap_recs + 1);
            const uint64_t need_cigar = std::max<uint64_t>(m_cig[R] + 16, reads->chunk_cap_cigar + 16);
            if (d_m_recs.n < need_recs) d_m_recs.alloc(grown(need_recs));
            if (d_m_cigar.n < need_cigar) d_m_cigar.alloc(grown(need_cigar));
        } else {
            d_m_aln.alloc(raw_pairs_cap + 1); d_m_cig.alloc(raw_pairs_cap + 1);
            d_m_recs.alloc(m_aln[R] + 1); d_m_cigar.alloc(m_cig[R] + 8); d_m_meta.alloc(raw_pairs_cap);
        }
        d_m_aln.upload(m_aln.data(), R + 1, s); d_m_cig.upload(m_cig.data(), R + 1, s);
        hipLaunchKernelGGL(merge_kernel, dim3(static_cast<uint32_t>(std::min<uint64_t>(R, 65535))), dim3(64), 0, s, reads->view(), d_new_cnt.p, d_new_words.p,
                           d_rec_at.p, d_word_at.p, xrecs, d_xwords.p, d_m_aln.p, d_m_cig.p, d_m_recs.p, d_m_cigar.p, d_m_meta.p);
        LCTY_HIP(hipGetLastError());
        LCTY_HIP(hipStreamSynchronize(s));
        std::swap(reads->d_aln_off, d_m_aln); std::swap(reads->d_cigar_off, d_m_cig);
        std::swap(reads->d_recs, d_m_recs); std::swap(reads->d_cigar, d_m_cigar); std::swap(reads->d_pair_meta, d_m_meta);
        reads->n_recs = m_aln[R]; reads->n_cigar = m_cig[R];
        if (keep) { reads->cap_recs = reads->d_recs.n - 1; reads->cap_cigar = reads->d_cigar.n - 16; }
        else { reads->cap_recs = reads->n_recs; reads->cap_cigar = reads->n_cigar; }           // the merged tables are exactly full
        // the pair-alignment arena was sized for the records of lcty_reads_create (same bound as there)
        uint64_t pa_cap = std::min<uint64_t>(static_cast<uint64_t>(LCTY_MAX_USED_ALNS) * reads->cap_pairs * A, 2 * reads->n_recs + reads->cap_pairs) + 64;
        if (reads->pa_pooled) pa_cap += pa_cap / 8 + static_cast<uint64_t>(PA_CHUNK) * PA_MAX_GRID;   // lcty_reads_create
        // (a streaming batch keeps the arena it was created with: the PairAlignments of the chunks before this one live in it)
        if (!reads->streaming && reads->d_pa.n < pa_cap) reads->d_pa.alloc(pa_cap);
        reads->max_recs_per_pair = std::max<uint32_t>(reads->streaming ? reads->max_recs_per_pair : 0u, static_cast<uint32_t>(max_recs));
        reads->max_cigar_per_pair = std::max<uint32_t>(reads->streaming ? reads->max_cigar_per_pair : 0u,
                                                       static_cast<uint32_t>(std::min<uint64_t>(max_cig, 0xFFFFFFF0ull)));
        reads->scored = false; reads->good_valid = false; reads->loc_table_valid = false;
        (void)total_words;
    });
}

int32_t lcty_recover_stats(lcty_reads* reads, uint64_t* level_pairs) {
    return guarded([&] {
        if (!reads || !level_pairs) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        for (int i = 0; i < 3; i++) level_pairs[i] = reads->recover_level_pairs[i];
    });
}

// cells of the aligner's dynamic-programming matrices filled by the last lcty_recover_alignments on this batch (GCUPS = cells / time)
int32_t lcty_recover_dp_cells(lcty_reads* reads, uint64_t* cells) {
    return guarded([&] {
        if (!reads || !cells) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        *cells = reads->recover_dp_cells;
    });
}

}  // extern "C"
