// lcty_map.hip — candidate generation inside a locus (first slice): where do the read ends of a chunk
// align on the BASIS alleles of the locus? The reference leaves this to an external mapper (strobealign for short reads,
// src/command/genotype.rs:962-1005; `| samtools view -e "[AS] >= 50 || flag & 2304 == 0"`, 1055-1094; mapping to the basis
// haplotypes only with --basis, 1007-1052) and reads its `aln.bam`; the other alleles are then reached through the
// haplotype-to-haplotype alignments (lcty_recover_alignments). No source of that mapper is in the reference tree: the
// algorithm here is this build's own, stated below and restated in tests/pyref_map.py, against which the kernel is bit-exact.
//
//   index    every k-mer (k <= 31, canonical form) of every basis allele -> its (basis allele, position, "forward is canonical"),
//            an open-addressing table of runs built on the host once per locus (basis sets are small: --basis).
//   seeds    the k-mers of a read end at every `stride`-th position plus the last one; k-mers with a base that is not ACGT are skipped,
//            and so are k-mers with more than max_occ places in the index (microsatellites; at most 64 places vote in any case).
//   votes    every (seed, index entry) pair votes for (basis allele, strand, diagonal). An (allele, strand) whose best diagonal has
//            >= half the votes of the read end's best one (strobealign -S 0.5) contributes its diagonals with >= min_votes votes and
//            >= half the votes of its best as candidates (a read end in a tandem repeat has as many votes one period off), the
//            first 64 in (allele, strand, diagonal) order.
//   extend   without gaps along the diagonal: +match per equal base, -mismatch otherwise (bases that are not ACGT never match),
//            `end_bonus` for each read end reached; the best-scoring stretch is the alignment, the rest is soft-clipped
//            ; per (allele, strand) the candidate with the best score stays (ties: the smallest diagonal).
//   gaps     a candidate that stayed and is clipped is aligned again WITH gaps: gap-affine (open + (n - 1) x extend for a gap of n) in
//            a band of +-band diagonals around its own, starting and ending on an aligned base, soft clips and end bonuses as above
//            (one lane = one candidate, the lanes of a wavefront taking candidates of any read ends from a list; M / deletion /
//            insertion scores of the running row in registers, the bases under the band in LDS, one direction nibble per cell in a
//            scratch of the workgroup, traceback into = / X / I / D runs). It replaces the alignment without gaps when
//            its score is higher. Order of preference on ties, fixed here and in the restatement: continue before starting
//            afresh, M before deletion before insertion as predecessor, open before extend, the first best end cell by (read
//            position, diagonal).
//   records  the best candidate of a read end is its primary record, the others with a score >= min_score are secondary
//            records (the samtools filter above); a read end without a candidate is an unmapped record. Record order, flags,
//            =/X/S CIGARs and SEQ orientation (reverse-complemented when the primary is on the reverse strand) are those of
//            the BAM the reference reads (model/locs.rs:1116-1150), so the chunk goes straight into lcty_reads_append.
//
// Three kernels: map_seed_kernel (one wavefront per read end; lane = seed, then lane = hit: bitonic sort of <= 1 024 vote keys in
// LDS, then lane = candidate, <= 64) leaves the candidates that stayed and lists the clipped ones; map_gap_kernel aligns the
// listed ones with gaps, one per lane; map_emit_kernel<WRITE> (one wavefront per read end, lane = candidate) runs twice: sizes,
// then records, with a host prefix sum in between.
#include <algorithm>
#include <chrono>

#include "lcty_map_internal.hpp"

namespace lcty {

namespace {

constexpr uint32_t MAP_MAX_HITS = 1024;
constexpr uint32_t MAP_PER_SEED = 64;      // index entries a seed may vote with (the first ones: by allele, then position)
constexpr uint32_t MAP_BAND_W = 2 * MAP_MAX_BAND + 1;
constexpr int32_t MAP_NEG = -(1 << 29);
constexpr uint32_t MAP_DIR_WORDS = (MAP_BAND_W + 7) / 8;                 // direction nibbles of a row, eight per word
constexpr uint32_t MAP_REFW = MAP_MAX_LEN + 2 * MAP_MAX_BAND;             // bases of the allele a band alignment can touch

// a candidate between the kernels
struct MapCand {
    uint32_t diag;                         // + 2^31
    int32_t score; uint16_t s, e;          // without gaps: score, aligned stretch [s, e) of the read end
    uint16_t g, state;                     // basis index * 2 + strand; 0: as extended without gaps, 1: listed for kernel 2, 2: aligned with gaps (below)
    int32_t g_score; uint32_t g_pos; uint16_t g_lead, g_trail; uint32_t ops_at; uint16_t g_inner, pad;
};

struct MapView {
    const MapSlot* table; uint64_t mask;
    const uint64_t* entries;               // basis index << 33 | position << 1 | forward-is-canonical
    const uint16_t* basis;                 // basis index -> allele
    uint32_t n_basis, k, stride, min_votes, max_occ, band;
    int32_t match, mismatch, end_bonus, min_score, gap_open, gap_extend;
    // between the kernels
    MapCand* cands; uint32_t slots;        // [read end][slots]: the candidates that stayed, in (allele, strand) order
    uint32_t* n_have;                      // per read end
    uint32_t* work; uint32_t n_work;       // slots of the candidates to be aligned with gaps
    uint32_t* counters;                    // [0] entries of `work`, [1] CIGAR words asked for in `ops`
    uint32_t* ops; uint32_t ops_cap;       // CIGAR words of the alignments with gaps (without the soft clips)
    uint32_t* scratch; uint32_t max_len;   // direction nibbles: [workgroup of kernel 2][max_len * MAP_DIR_WORDS][64]
    const uint8_t* seqs; const uint64_t* seq_off; const uint32_t* allele_len;
    // reads
    uint64_t n_mates;
    const uint32_t* mate_len; const uint64_t* mate_off; const uint32_t* bases2; const uint32_t* nmask;
    int paired;
    // pass 1
    uint32_t* n_recs; uint32_t* n_cigar;
    // pass 2
    const uint64_t* rec_at; const uint64_t* cig_at;      // per read end: first record, first CIGAR word
    const uint64_t* pair_cig;                            // per pair: first CIGAR word (records carry offsets relative to it)
    lcty_aln_rec* recs; uint32_t* cigar;
    uint32_t* out_bases2; uint32_t* out_nmask;
};

// ---- kernel 1: seeds -> votes -> candidates -> extension without gaps; the candidates that stay, in (allele, strand) order
__device__ void map_seed_one(const MapView& V, const uint64_t m, uint64_t* keys, unsigned long long* best, uint64_t* cand_key) {
    const uint32_t lane = threadIdx.x;
    const uint32_t L = V.mate_len[m];
    if (L == 0) {                                                               // absent read end (single-end data)
        if (lane == 0) V.n_have[m] = 0;
        return;
    }
    const uint64_t off = V.mate_off[m];
    const uint32_t k = V.k;
    // ---- seeds and votes
    uint32_t n_hits = 0;
    if (L >= k) {
        const uint32_t span = L - k, n0 = span / V.stride + 1, n_seeds = n0 + (span % V.stride ? 1u : 0u);
        uint32_t start = 0, count = 0, pr = 0;
        bool read_fwd = false;
        if (lane < n_seeds) {
            pr = lane < n0 ? lane * V.stride : span;
            uint64_t fw = 0, rv = 0;
            bool bad = false;
            for (uint32_t j = 0; j < k; j++) {
                const uint32_t e = base_at(V.bases2, off, pr + j);
                bad |= n_at(V.nmask, off, pr + j);
                fw = (fw << 2) | e;
                rv = (rv >> 2) | (static_cast<uint64_t>(3u - e) << (2 * k - 2));
            }
            if (!bad) {
                read_fwd = fw <= rv;
                const uint64_t canon = read_fwd ? fw : rv;
                uint64_t h = map_hash(canon) & V.mask;
                for (;;) {
                    const MapSlot s = V.table[h];
                    if (s.key == MAP_FREE) break;
                    if (s.key == canon) { start = s.start; count = s.count > V.max_occ ? 0u : min(s.count, MAP_PER_SEED); break; }      // a repetitive k-mer says nothing about the place
                    h = (h + 1) & V.mask;
                }
            }
        }
        // places of the lanes' votes in the list (seed order); what does not fit is dropped
        uint32_t incl = count;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(incl, o); if (lane >= static_cast<uint32_t>(o)) incl += up; }
        const uint32_t at = incl - count;
        n_hits = min(static_cast<uint32_t>(__shfl(incl, 63)), MAP_MAX_HITS);
        for (uint32_t j = 0; j < count && at + j < MAP_MAX_HITS; j++) {
            const uint64_t e = V.entries[start + j];
            const uint32_t b = static_cast<uint32_t>(e >> 33), pa = static_cast<uint32_t>(e >> 1);
            const bool allele_fwd = e & 1ull;
            const uint32_t strand = read_fwd == allele_fwd ? 0u : 1u;
            const int64_t diag = strand ? static_cast<int64_t>(pa) - static_cast<int64_t>(L - k - pr) : static_cast<int64_t>(pa) - static_cast<int64_t>(pr);
            keys[at + j] = (static_cast<uint64_t>(b * 2 + strand) << 32) | static_cast<uint32_t>(diag + 0x80000000ll);
        }
    }
    // ---- sort the votes (bitonic over the next power of two, padded with the largest key)
    uint32_t n2 = 1;
    while (n2 < n_hits) n2 <<= 1;
    for (uint32_t i = n_hits + lane; i < n2; i += 64) keys[i] = ~0ull;
    for (uint32_t i = lane; i < 2 * MAP_MAX_BASIS; i += 64) best[i] = 0ull;
    __syncthreads();
    for (uint32_t size = 2; size <= n2; size <<= 1)
        for (uint32_t strd = size >> 1; strd > 0; strd >>= 1) {
            for (uint32_t i = lane; i < n2 / 2; i += 64) {
                const uint32_t lo = 2 * i - (i & (strd - 1)), hi = lo + strd;
                const bool up = (lo & size) == 0;
                const uint64_t a = keys[lo], b = keys[hi];
                if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
            }
            __syncthreads();
        }
    // ---- per (allele, strand): the votes of its best diagonal
    for (uint32_t i = lane; i < n_hits; i += 64) {
        const uint64_t key = keys[i];
        if (i > 0 && keys[i - 1] == key) continue;
        uint32_t len = 1;
        while (i + len < n_hits && keys[i + len] == key) len++;
        atomicMax(&best[key >> 32], static_cast<unsigned long long>(len));
    }
    __syncthreads();
    const uint32_t n_groups = 2 * V.n_basis;
    uint32_t max_votes = lane < n_groups ? static_cast<uint32_t>(best[lane]) : 0u;
    for (int o = 32; o > 0; o >>= 1) max_votes = max(max_votes, static_cast<uint32_t>(__shfl_xor(static_cast<int>(max_votes), o)));
    // ---- candidates: the diagonals of an (allele, strand) with >= min_votes votes and >= half the votes of its best one (a read end
    // in a tandem repeat has as many votes one period off: the extension decides), for the (allele, strand)s whose best diagonal has
    // >= half the votes of the read end's best; in (allele, strand, diagonal) order, the first 64
    uint32_t n_cand = 0;
    for (uint32_t i0 = 0; i0 < n_hits && n_cand < 64; i0 += 64) {
        const uint32_t i = i0 + lane;
        bool q = false;
        if (i < n_hits) {
            const uint64_t key = keys[i];
            if (i == 0 || keys[i - 1] != key) {
                uint32_t len = 1;
                while (i + len < n_hits && keys[i + len] == key) len++;
                const uint32_t gv = static_cast<uint32_t>(best[key >> 32]);
                q = len >= V.min_votes && 2 * len >= gv && 2 * gv >= max_votes;
            }
        }
        const unsigned long long qm = __ballot(q);
        const uint32_t at = n_cand + static_cast<uint32_t>(__popcll(qm & ((1ull << lane) - 1ull)));
        if (q && at < 64) cand_key[at] = keys[i];
        n_cand = min(64u, n_cand + static_cast<uint32_t>(__popcll(qm)));
    }
    __syncthreads();
    // ---- extension: lane = candidate
    const bool cand = lane < n_cand;
    const uint64_t ckey = cand ? cand_key[lane] : 0ull;
    const uint32_t g = static_cast<uint32_t>(ckey >> 32);
    const uint32_t strand = g & 1u, allele = cand ? V.basis[g >> 1] : 0u;
    const int64_t diag = static_cast<int64_t>(static_cast<uint32_t>(ckey)) - 0x80000000ll;
    const uint8_t* ref = V.seqs + V.seq_off[allele];
    const int64_t alen = V.allele_len[allele];
    // the best-scoring stretch [s, e) inside the part of the read end that lies on the allele
    int32_t score = INT32_MIN; uint32_t s_best = 0, e_best = 0;
    if (cand) {
        const uint32_t i_lo = diag < 0 ? static_cast<uint32_t>(-diag) : 0u;
        const uint32_t i_hi = static_cast<uint32_t>(max<int64_t>(0, min<int64_t>(L, alen - diag)));
        int32_t run = 0; uint32_t run_s = 0; bool open = false;
        for (uint32_t i = i_lo; i < i_hi; i++) {
            const int32_t fresh = i == 0 ? V.end_bonus : 0;
            if (!open || fresh > run) { run = fresh; run_s = i; open = true; }   // a stretch that starts here (an older start wins ties)
            const uint32_t src = strand ? L - 1 - i : i;
            const uint32_t e = strand ? 3u - base_at(V.bases2, off, src) : base_at(V.bases2, off, src);
            run += !n_at(V.nmask, off, src) && enc_of(ref[diag + i]) == e ? V.match : -V.mismatch;
            const int32_t total = run + (i + 1 == L ? V.end_bonus : 0);
            if (total > score) { score = total; s_best = run_s; e_best = i + 1; }
        }
    }
    // the best diagonal of every (allele, strand): highest score, the smallest diagonal (= first lane) on ties
    for (uint32_t i = lane; i < 2 * MAP_MAX_BASIS; i += 64) best[i] = 0ull;
    __syncthreads();
    if (cand && score > INT32_MIN)
        atomicMax(&best[g], (static_cast<unsigned long long>(static_cast<uint32_t>(score) ^ 0x80000000u) << 32) | (63u - lane));
    __syncthreads();
    const bool have = cand && score > INT32_MIN && static_cast<uint32_t>(best[g]) == 63u - lane;
    // a clipped candidate that stayed goes to the list of kernel 2
    const bool need = have && V.band > 0 && (s_best > 0 || e_best < L);
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long hm = __ballot(have), wm = __ballot(need);
    const uint32_t slot = static_cast<uint32_t>(m) * V.slots + static_cast<uint32_t>(__popcll(hm & below));
    if (have) {
        MapCand c{};
        c.diag = static_cast<uint32_t>(ckey); c.score = score; c.s = static_cast<uint16_t>(s_best); c.e = static_cast<uint16_t>(e_best);
        c.g = static_cast<uint16_t>(g); c.state = need ? 1u : 0u;
        V.cands[slot] = c;
    }
    if (lane == 0) V.n_have[m] = static_cast<uint32_t>(__popcll(hm));
    if (wm) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&V.counters[0], static_cast<uint32_t>(__popcll(wm)));
        base = static_cast<uint32_t>(__shfl(static_cast<int>(base), 0));
        if (need) V.work[base + static_cast<uint32_t>(__popcll(wm & below))] = slot;
    }
}

__global__ __launch_bounds__(64) void map_seed_kernel(const MapView V) {
    __shared__ uint64_t keys[MAP_MAX_HITS];
    __shared__ unsigned long long best[2 * MAP_MAX_BASIS];
    __shared__ uint64_t cand_key[64];
    for (uint64_t m = blockIdx.x; m < V.n_mates; m += gridDim.x) {
        map_seed_one(V, m, keys, best, cand_key);
        __syncthreads();
    }
}

// ---- kernel 2: the clipped candidates again, with gaps (the header comment states the recurrence and the tie rules). One lane = one
// candidate of the list, whichever read end it belongs to: every lane of a wavefront runs a band alignment of its own.
__global__ __launch_bounds__(64) void map_gap_kernel(const MapView V) {
    __shared__ uint8_t wantw[MAP_MAX_LEN * 64];         // [i][lane]: the allele base that equals base i of the read end in alignment orientation (6: none does)
    __shared__ uint8_t refw[MAP_REFW * 64];             // [j][lane]: the allele base at diag - band + j (4: not ACGT, 5: outside the allele)
    const uint32_t lane = threadIdx.x;
    uint32_t* dirs = V.scratch + static_cast<size_t>(blockIdx.x) * V.max_len * MAP_DIR_WORDS * 64 + lane;      // [(i * MAP_DIR_WORDS + word) * 64]
    const int32_t B = static_cast<int32_t>(V.band), W = 2 * B + 1;
    for (uint32_t w0 = blockIdx.x * 64u; w0 < V.n_work; w0 += gridDim.x * 64u) {
        if (w0 + lane >= V.n_work) continue;            // the columns of LDS are the lanes' own: no barrier below
        const uint32_t slot = V.work[w0 + lane];
        MapCand c = V.cands[slot];
        const uint64_t m = slot / V.slots;
        const uint32_t L = V.mate_len[m];
        const uint64_t off = V.mate_off[m];
        const uint32_t strand = c.g & 1u, allele = V.basis[c.g >> 1];
        const int64_t diag = static_cast<int64_t>(c.diag) - 0x80000000ll;
        const uint8_t* ref = V.seqs + V.seq_off[allele];
        const int64_t alen = V.allele_len[allele];
        for (uint32_t i = 0; i < L; i++) {
            const uint32_t src = strand ? L - 1 - i : i;
            const uint32_t r = base_at(V.bases2, off, src);
            wantw[i * 64 + lane] = n_at(V.nmask, off, src) ? 6u : static_cast<uint8_t>(strand ? 3u - r : r);
        }
        const uint32_t span = L + 2 * static_cast<uint32_t>(B);
        for (uint32_t j = 0; j < span; j++) {
            const int64_t rp = diag - B + static_cast<int64_t>(j);
            refw[j * 64 + lane] = rp >= 0 && rp < alen ? static_cast<uint8_t>(enc_of(ref[rp])) : 5u;
        }
        // the running row of M / deletion / insertion scores in registers: the loop over the band is unrolled to its full width
        // (a narrower band leaves the outer diagonals at "no alignment")
        int32_t M[MAP_BAND_W], E[MAP_BAND_W], F[MAP_BAND_W];
#pragma unroll
        for (uint32_t k = 0; k < MAP_BAND_W; k++) { M[k] = MAP_NEG; E[k] = MAP_NEG; F[k] = MAP_NEG; }
        int32_t best_total = INT32_MIN; uint32_t end_i = 0; int32_t end_k = 0;
        for (uint32_t i = 0; i < L; i++) {
            int32_t left_m = MAP_NEG, left_e = MAP_NEG;
            const int32_t fresh = i == 0 ? V.end_bonus : 0;
            const uint32_t want = wantw[i * 64 + lane];
            uint32_t packed = 0;
#pragma unroll
            for (uint32_t k = 0; k < MAP_BAND_W; k++) {
                if (static_cast<int32_t>(k) < W) {
                    const uint32_t rbase = refw[(i + k) * 64 + lane];        // the allele position diag + i + (k - B) is entry i + k of the window
                    const bool inref = rbase != 5u;
                    const int32_t om = M[k], oe = E[k], of = F[k];
                    const int32_t rm = k + 1 < MAP_BAND_W && static_cast<int32_t>(k) + 1 < W ? M[k + 1 < MAP_BAND_W ? k + 1 : k] : MAP_NEG;
                    const int32_t rf = k + 1 < MAP_BAND_W && static_cast<int32_t>(k) + 1 < W ? F[k + 1 < MAP_BAND_W ? k + 1 : k] : MAP_NEG;
                    int32_t prev = om; uint32_t code = 1;
                    if (oe > prev) { prev = oe; code = 2; }
                    if (of > prev) { prev = of; code = 3; }
                    if (fresh > prev) { prev = fresh; code = 0; }
                    int32_t nm = inref ? prev + (rbase == want ? V.match : -V.mismatch) : MAP_NEG;
                    const int32_t fo = rm - V.gap_open, fe = rf - V.gap_extend;
                    int32_t nf = fe > fo ? fe : fo; const uint32_t fcode = fe > fo ? 1u : 0u;
                    const int32_t eo = left_m - V.gap_open, ee = left_e - V.gap_extend;
                    int32_t ne = ee > eo ? ee : eo; const uint32_t ecode = ee > eo ? 1u : 0u;
                    if (!inref) ne = MAP_NEG;
                    if (nm < MAP_NEG / 2) nm = MAP_NEG;
                    if (nf < MAP_NEG / 2) nf = MAP_NEG;
                    if (ne < MAP_NEG / 2) ne = MAP_NEG;
                    packed |= (code | (ecode << 2) | (fcode << 3)) << (4 * (k & 7));
                    if ((k & 7) == 7 || static_cast<int32_t>(k) + 1 == W) { dirs[static_cast<size_t>(i * MAP_DIR_WORDS + (k >> 3)) * 64] = packed; packed = 0; }
                    M[k] = nm; E[k] = ne; F[k] = nf;
                    left_m = nm; left_e = ne;
                    if (nm > MAP_NEG) {
                        const int32_t total = nm + (i + 1 == L ? V.end_bonus : 0);
                        if (total > best_total) { best_total = total; end_i = i; end_k = static_cast<int32_t>(k); }
                    }
                }
            }
        }
        if (best_total <= c.score) continue;
        // traceback, twice: the number of CIGAR words, then the words (met last first) into their place of the list
        uint32_t n_words = 0, at = 0, lead = 0; int32_t k_first = 0;
        for (int pass = 0; pass < 2; pass++) {
            uint32_t i = end_i; int32_t k = end_k; uint32_t state = 0, n = 0, cur_op = 0xFFFFFFFFu, cur_len = 0;
            auto emit = [&](uint32_t op) {
                if (op == cur_op) { cur_len++; return; }
                if (cur_len) { if (pass) V.ops[at + n_words - 1 - n] = (cur_len << 4) | cur_op; n++; }
                cur_op = op; cur_len = 1;
            };
            for (;;) {
                const uint32_t d = (dirs[static_cast<size_t>(i * MAP_DIR_WORDS + (static_cast<uint32_t>(k) >> 3)) * 64] >> (4 * (k & 7))) & 15u;
                if (state == 0) {
                    emit(refw[(i + static_cast<uint32_t>(k)) * 64 + lane] == wantw[i * 64 + lane] ? 7u : 8u);
                    const uint32_t cc = d & 3u;
                    if (cc == 0) break;
                    state = cc - 1;                                             // 1 -> M, 2 -> deletion, 3 -> insertion; all at (i - 1, k)
                    i--;
                } else if (state == 1) {
                    emit(2u);                                                   // D
                    state = (d >> 2) & 1u ? 1u : 0u;
                    k--;
                } else {
                    emit(1u);                                                   // I
                    state = (d >> 3) & 1u ? 2u : 0u;
                    i--; k++;
                }
            }
            if (cur_len) { if (pass) V.ops[at + n_words - 1 - n] = (cur_len << 4) | cur_op; n++; }
            if (!pass) {
                n_words = n; lead = i; k_first = k;
                at = atomicAdd(&V.counters[1], n_words);
                if (at + n_words > V.ops_cap || at + n_words < at) break;      // the host repeats the kernel with room for counters[1] words
            } else {
                c.state = 2; c.g_score = best_total; c.g_pos = static_cast<uint32_t>(diag + static_cast<int64_t>(lead) + (k_first - B));
                c.g_lead = static_cast<uint16_t>(lead); c.g_trail = static_cast<uint16_t>(L - 1 - end_i); c.g_inner = static_cast<uint16_t>(n_words); c.ops_at = at;
                V.cands[slot] = c;
            }
        }
    }
}

// ---- kernel 3: the records of a read end from its candidates; sizes (WRITE = false), then the records themselves
template <bool WRITE>
__device__ void map_emit_one(const MapView& V, const uint64_t m) {
    const uint32_t lane = threadIdx.x;
    const uint32_t L = V.mate_len[m];
    if (L == 0) {
        if (!WRITE && lane == 0) { V.n_recs[m] = 0; V.n_cigar[m] = 0; }
        return;
    }
    const uint64_t off = V.mate_off[m];
    const bool have = lane < V.n_have[m];
    MapCand c{};
    if (have) c = V.cands[static_cast<uint32_t>(m) * V.slots + lane];
    const bool gapped = have && c.state == 2;
    const int32_t score = gapped ? c.g_score : c.score;
    const uint32_t strand = c.g & 1u, allele = have ? V.basis[c.g >> 1] : 0u;
    const int64_t diag = static_cast<int64_t>(c.diag) - 0x80000000ll;
    const uint8_t* ref = V.seqs + V.seq_off[allele];
    const uint32_t s_best = c.s, e_best = c.e;
    // a position of the read end in alignment orientation: equal to the allele's base there?
    auto equal_at = [&](uint32_t i) -> bool {
        const uint32_t src = strand ? L - 1 - i : i;
        if (n_at(V.nmask, off, src)) return false;
        const uint32_t e = strand ? 3u - base_at(V.bases2, off, src) : base_at(V.bases2, off, src);
        return enc_of(ref[diag + i]) == e;
    };
    // ---- the primary record: best score, the smallest (allele, strand) on ties (the lanes are in that order)
    int32_t top = have ? score : INT32_MIN;
    for (int o = 32; o > 0; o >>= 1) top = max(top, __shfl_xor(top, o));
    const unsigned long long tops = __ballot(have && score == top);
    const uint32_t lp = tops ? static_cast<uint32_t>(__ffsll(static_cast<long long>(tops))) - 1u : 0xFFFFFFFFu;
    const bool keep = have && (lane == lp || score >= V.min_score);
    // CIGAR words of a kept candidate: [S] runs of = / X [S], or what kernel 2 left
    uint32_t n_ops = 0;
    if (keep && gapped) n_ops = c.g_inner + (c.g_lead > 0) + (c.g_trail > 0);
    else if (keep) {
        n_ops = (s_best > 0) + (e_best < L);
        bool prev = false;
        for (uint32_t i = s_best; i < e_best; i++) { const bool eq = equal_at(i); if (i == s_best || eq != prev) n_ops++; prev = eq; }
    }
    const unsigned long long kept = __ballot(keep);
    const uint32_t n_kept = static_cast<uint32_t>(__popcll(kept));
    uint32_t ops_incl = keep && lane != lp ? n_ops : 0u;                         // the others follow the primary in (allele, strand) order
    for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(ops_incl, o); if (lane >= static_cast<uint32_t>(o)) ops_incl += up; }
    const uint32_t ops_primary = lp != 0xFFFFFFFFu ? static_cast<uint32_t>(__shfl(static_cast<int>(n_ops), static_cast<int>(lp))) : 0u;
    const uint32_t ops_total = ops_primary + static_cast<uint32_t>(__shfl(static_cast<int>(ops_incl), 63));
    if (!WRITE) {
        if (lane == 0) { V.n_recs[m] = n_kept ? n_kept : 1u; V.n_cigar[m] = ops_total; }       // no candidate: one unmapped record
        uint32_t widest = n_ops;
        for (int o = 32; o > 0; o >>= 1) widest = max(widest, static_cast<uint32_t>(__shfl_xor(static_cast<int>(widest), o)));
        if (lane == 0 && widest > V.counters[2]) atomicMax(&V.counters[2], widest);
        return;
    }
    const uint32_t mate2 = V.paired && (m & 1u) ? LCTY_FLAG_MATE2 : 0u;
    const uint64_t rec0 = V.rec_at[m], cig0 = V.cig_at[m], rel0 = cig0 - V.pair_cig[m >> 1];
    const bool primary_reverse = lp != 0xFFFFFFFFu && __shfl(static_cast<int>(strand), static_cast<int>(lp)) != 0;
    if (n_kept == 0) {
        if (lane == 0) V.recs[rec0] = lcty_aln_rec{0u, 0u, static_cast<uint16_t>(LCTY_FLAG_UNMAPPED | mate2), 0u, static_cast<uint32_t>(rel0)};
    } else if (keep) {
        const uint32_t rank = lane == lp ? 0u : 1u + static_cast<uint32_t>(__popcll(kept & ((1ull << lane) - 1ull) & ~(1ull << lp)));
        const uint32_t cig_rel = lane == lp ? 0u : ops_primary + ops_incl - n_ops;
        uint32_t* cg = V.cigar + cig0 + cig_rel;
        uint32_t w = 0;
        if (gapped) {
            if (c.g_lead > 0) cg[w++] = (static_cast<uint32_t>(c.g_lead) << 4) | 4u;
            for (uint32_t j = 0; j < c.g_inner; j++) cg[w++] = V.ops[c.ops_at + j];
            if (c.g_trail > 0) cg[w++] = (static_cast<uint32_t>(c.g_trail) << 4) | 4u;
        } else {
            if (s_best > 0) cg[w++] = (s_best << 4) | 4u;                        // S
            bool prev = false; uint32_t len = 0;
            for (uint32_t i = s_best; i < e_best; i++) {
                const bool eq = equal_at(i);
                if (i > s_best && eq != prev) { cg[w++] = (len << 4) | (prev ? 7u : 8u); len = 0; }      // = / X
                prev = eq; len++;
            }
            cg[w++] = (len << 4) | (prev ? 7u : 8u);
            if (e_best < L) cg[w++] = ((L - e_best) << 4) | 4u;
        }
        const uint16_t flags = static_cast<uint16_t>((strand ? LCTY_FLAG_REVERSE : 0u) | (lane == lp ? 0u : LCTY_FLAG_SECONDARY) | mate2);
        V.recs[rec0 + rank] = lcty_aln_rec{gapped ? c.g_pos : static_cast<uint32_t>(diag + s_best), static_cast<uint16_t>(allele), flags, n_ops,
                                           static_cast<uint32_t>(rel0 + cig_rel)};
    }
    // SEQ as the BAM has it: reverse-complemented when the primary record is on the reverse strand. The read end owns whole
    // 32-base words of the output (offsets are multiples of 32); lane = output word of 16 bases.
    const uint32_t words = (L + 15) / 16;
    for (uint32_t wi = lane; wi < words; wi += 64) {
        uint32_t out = 0;
        for (uint32_t j = 0; j < 16 && wi * 16 + j < L; j++) {
            const uint32_t i = wi * 16 + j, src = primary_reverse ? L - 1 - i : i;
            const uint32_t e = base_at(V.bases2, off, src);
            out |= (primary_reverse ? 3u - e : e) << (2 * j);
        }
        V.out_bases2[(off >> 4) + wi] = out;
    }
    for (uint32_t wi = lane; wi < (L + 31) / 32; wi += 64) {
        uint32_t out = 0;
        for (uint32_t j = 0; j < 32 && wi * 32 + j < L; j++) {
            const uint32_t i = wi * 32 + j, src = primary_reverse ? L - 1 - i : i;
            out |= static_cast<uint32_t>(n_at(V.nmask, off, src)) << j;
        }
        V.out_nmask[(off >> 5) + wi] = out;
    }
}

template <bool WRITE>
__global__ __launch_bounds__(64) void map_emit_kernel(const MapView V) {
    for (uint64_t m = blockIdx.x; m < V.n_mates; m += gridDim.x) map_emit_one<WRITE>(V, m);
}

}  // namespace

}  // namespace lcty

using namespace lcty;

extern "C" {

int32_t lcty_map_params_default(lcty_map_params* p) {
    return guarded([&] {
        if (!p) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        memset(p, 0, sizeof(*p));
        p->k = 15;                 // strobealign -k 15 in the reference's command line (genotype.rs:984)
        p->stride = 5;
        p->min_votes = 2;
        p->max_occ = 0;            // 0: four times the number of basis alleles (strobealign -f / minimap2 -f: the most repetitive seeds are left out)
        p->match = 2; p->mismatch = 8; p->end_bonus = 10;      // strobealign's scores
        p->band = 16; p->gap_open = 12; p->gap_extend = 1;     // a gap of n bases costs gap_open + (n - 1) * gap_extend; band 0: no alignment with gaps
        p->min_score = 50;         // samtools view -e "[AS] >= 50 || flag & 2304 == 0" (genotype.rs:1070)
        p->route = LCTY_MAP_ROUTE_AUTO;
        p->chain_gap = 2000; p->chain_skew = 500; p->chain_back = 32;      // long route only
    });
}

int32_t lcty_map_params_default_long(lcty_map_params* p) {
    return guarded([&] {
        if (!p) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        memset(p, 0, sizeof(*p));
        p->k = 15; p->stride = 16;                               // minimap2 map-ont / map-hifi seeds are 15..19 bases, one in ~5..10 kept
        p->min_votes = 3;                                        // minimap2 -n 3: anchors in a chain
        p->max_occ = 0;
        p->match = 2; p->mismatch = 4; p->end_bonus = 10;        // minimap2 -A 2 -B 4
        p->band = 16; p->gap_open = 6; p->gap_extend = 2;        // -O 4 -E 2: a gap of n bases costs 4 + 2 n
        p->min_score = INT32_MIN;                                // no samtools filter for long reads (genotype.rs:1069-1071)
        p->route = LCTY_MAP_ROUTE_AUTO;
        p->chain_gap = 2000; p->chain_skew = 500; p->chain_back = 32;
    });
}

int32_t lcty_locus_build_map_index(lcty_locus* locus, const uint16_t* basis, uint32_t n_basis, uint32_t k) {
    return guarded([&] {
        if (!locus || !basis) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (n_basis == 0 || n_basis > MAP_LONG_MAX_BASIS) fail(LCTY_ERR_UNSUPPORTED, "1..%u basis alleles", MAP_LONG_MAX_BASIS);
        if (k < 8 || k > 31) fail(LCTY_ERR_UNSUPPORTED, "seed length %u: 8..31", k);
        lcty_ctx* ctx = locus->ctx;
        ctx->activate();
        for (uint32_t b = 0; b < n_basis; b++)
            if (basis[b] >= locus->n_alleles) fail(LCTY_ERR_INVALID_INPUT, "basis allele %u >= %u", basis[b], locus->n_alleles);
        auto ix = build_map_index_device(locus, basis, n_basis, k);             // lcty_map_index.hip
        locus->map_index = ix;
    });
}

namespace {

void run_map(lcty_locus* locus, const lcty_reads_host* chunk, const lcty_map_params* params, uint64_t* aln_off, uint64_t* cigar_off, bool sizes_only,
             MapRun& X) {
    if (!locus || !chunk || !params || !aln_off || !cigar_off) fail(LCTY_ERR_INVALID_INPUT, "null argument");
    auto ix = std::static_pointer_cast<MapIndex>(locus->map_index);
    if (!ix) fail(LCTY_ERR_INVALID_INPUT, "lcty_locus_build_map_index has not been called on this locus");
    if (params->k != ix->k) fail(LCTY_ERR_INVALID_INPUT, "the index was built for k = %u", ix->k);
    if (params->stride == 0 || params->match <= 0 || params->mismatch < 0 || params->end_bonus < 0 || params->gap_open < 0 || params->gap_extend < 0)
        fail(LCTY_ERR_INVALID_INPUT, "mapper parameters");
    if (params->band > MAP_MAX_BAND) fail(LCTY_ERR_UNSUPPORTED, "band of at most %u diagonals on either side", MAP_MAX_BAND);
    lcty_ctx* ctx = locus->ctx;
    ctx->activate();
    hipStream_t s = ctx->stream;
    const uint64_t n = chunk->n_pairs, n_mates = 2 * n;
    aln_off[0] = 0; cigar_off[0] = 0;
    if (n == 0) return;
    const uint64_t nb = chunk->mate_off[n_mates];
    uint32_t max_len = 1;
    for (uint64_t m = 0; m < n_mates; m++) {
        max_len = std::max(max_len, chunk->mate_len[m]);
        if (chunk->mate_off[m] % 32) fail(LCTY_ERR_INVALID_INPUT, "mate offsets must be multiples of 32 bases");
    }
    // the route: read ends of up to 256 bases on up to 32 basis alleles vote for diagonals (this file); anything else is chained
    // and aligned along the chain (lcty_map_long.hip)
    if (params->route > LCTY_MAP_ROUTE_LONG) fail(LCTY_ERR_INVALID_INPUT, "mapper route %u", params->route);
    const bool fits_short = max_len <= MAP_MAX_LEN && ix->n_basis <= MAP_MAX_BASIS;
    const bool long_route = params->route == LCTY_MAP_ROUTE_LONG || (params->route == LCTY_MAP_ROUTE_AUTO && !fits_short);
    if (!long_route) {
        if (ix->n_basis > MAP_MAX_BASIS) fail(LCTY_ERR_UNSUPPORTED, "up to %u basis alleles on the short route (the index has %u)", MAP_MAX_BASIS, ix->n_basis);
        for (uint64_t m = 0; m < n_mates; m++) {
            if (chunk->mate_len[m] > MAP_MAX_LEN) fail(LCTY_ERR_UNSUPPORTED, "read ends of up to %u bases on the short route (this one: %u)", MAP_MAX_LEN, chunk->mate_len[m]);
            if (chunk->mate_len[m] >= params->k && (chunk->mate_len[m] - params->k) / params->stride + 2 > 64)
                fail(LCTY_ERR_UNSUPPORTED, "more than 64 seeds per read end: raise the stride");
        }
    }
    X.d_len.ensure_slack(n_mates); X.d_len.upload(chunk->mate_len, n_mates, s);
    X.d_off.ensure_slack(n_mates + 1); X.d_off.upload(chunk->mate_off, n_mates + 1, s);
    X.d_b2.ensure_slack(std::max<uint64_t>(nb / 16, 1)); X.d_b2.upload(chunk->bases2, nb / 16, s);
    X.d_nm.ensure_slack(std::max<uint64_t>(nb / 32, 1)); X.d_nm.upload(chunk->nmask, nb / 32, s);
    X.d_nrec.ensure_slack(n_mates); X.d_ncig.ensure_slack(n_mates);
    if (long_route) {
        run_map_long(locus, chunk, params, *ix, max_len, aln_off, cigar_off, sizes_only, X);
        return;
    }
    MapView V{};
    V.table = ix->table.p; V.mask = ix->mask; V.entries = ix->entries.p; V.basis = ix->basis.p; V.n_basis = ix->n_basis;
    V.k = params->k; V.stride = params->stride; V.min_votes = std::max<uint32_t>(params->min_votes, 1);
    V.max_occ = params->max_occ ? params->max_occ : 4 * ix->n_basis;
    V.match = params->match; V.mismatch = params->mismatch; V.end_bonus = params->end_bonus; V.min_score = params->min_score;
    V.band = params->band; V.gap_open = params->gap_open; V.gap_extend = params->gap_extend;
    const uint32_t cus = static_cast<uint32_t>(ctx->props.multiProcessorCount);
    const uint32_t n_wg = static_cast<uint32_t>(std::min<uint64_t>(n_mates, 16ull * cus));
    V.seqs = locus->d_seqs.p; V.seq_off = locus->d_seq_off.p; V.allele_len = locus->d_allele_len.p;
    V.n_mates = n_mates; V.mate_len = X.d_len.p; V.mate_off = X.d_off.p; V.bases2 = X.d_b2.p; V.nmask = X.d_nm.p;
    V.paired = locus->bg.is_paired;
    V.n_recs = X.d_nrec.p; V.n_cigar = X.d_ncig.p;
    // kernel 1: the candidates that stay, and the list of those to be aligned with gaps
    V.slots = std::min<uint32_t>(64, 2 * ix->n_basis);
    if (n_mates * V.slots > 0xFFFFFFFFull) fail(LCTY_ERR_UNSUPPORTED, "chunks of up to %llu read pairs with this basis", (unsigned long long)(0xFFFFFFFFull / V.slots / 2));
    X.d_cands.ensure_slack(n_mates * V.slots * sizeof(MapCand)); X.d_nhave.ensure_slack(n_mates); X.d_work.ensure_slack(n_mates * V.slots);
    X.d_counters.ensure_slack(4); X.d_counters.zero(s);
    V.cands = reinterpret_cast<MapCand*>(X.d_cands.p); V.n_have = X.d_nhave.p; V.work = X.d_work.p; V.counters = X.d_counters.p;
    ctx->timed(LCTY_K_MAP, [&] { hipLaunchKernelGGL(map_seed_kernel, dim3(n_wg), dim3(64), 0, s, V); }, s);
    LCTY_HIP(hipGetLastError());
    uint32_t counters[4] = {0, 0, 0, 0};
    X.d_counters.download(counters, 4, s);
    LCTY_HIP(hipStreamSynchronize(s));
    // kernel 2: the band alignments, repeated with more room if their CIGAR words did not fit
    V.n_work = counters[0];
    if (V.n_work) {
        const uint32_t n_wg2 = static_cast<uint32_t>(std::min<uint64_t>((V.n_work + 63) / 64, 4ull * cus));
        V.max_len = max_len;
        ix->scratch.ensure(static_cast<size_t>(n_wg2) * max_len * MAP_DIR_WORDS * 64);
        V.scratch = ix->scratch.p;
        uint64_t cap = 6ull * V.n_work + 1024;
        for (;;) {
            if (cap > 0xFFFFFFF0ull) fail(LCTY_ERR_UNSUPPORTED, "CIGAR words of the alignments with gaps: map the chunk in parts");
            X.d_ops.ensure_slack(cap);
            V.ops = X.d_ops.p; V.ops_cap = static_cast<uint32_t>(cap);
            ctx->timed(LCTY_K_MAP, [&] { hipLaunchKernelGGL(map_gap_kernel, dim3(n_wg2), dim3(64), 0, s, V); }, s);
            LCTY_HIP(hipGetLastError());
            X.d_counters.download(counters, 4, s);
            LCTY_HIP(hipStreamSynchronize(s));
            if (counters[1] <= cap) break;
            cap = static_cast<uint64_t>(counters[1]) + 1024;
            const uint32_t zero = 0;
            X.d_counters.upload(&zero, 1, s, 1);
        }
    }
    // kernel 3, sizes
    ctx->timed(LCTY_K_MAP, [&] { hipLaunchKernelGGL(map_emit_kernel<false>, dim3(n_wg), dim3(64), 0, s, V); }, s);
    LCTY_HIP(hipGetLastError());
    X.d_counters.download(counters, 4, s);
    X.nrec.resize(n_mates); X.ncig.resize(n_mates);
    X.d_nrec.download(X.nrec.data(), n_mates, s); X.d_ncig.download(X.ncig.data(), n_mates, s);
    LCTY_HIP(hipStreamSynchronize(s));
    X.max_rec_cigar = counters[2];
    std::vector<uint64_t> rec_at(n_mates), cig_at(n_mates), pair_cig(n);
    uint64_t r = 0, c = 0;
    for (uint64_t p = 0; p < n; p++) {
        pair_cig[p] = c;
        for (uint32_t e = 0; e < 2; e++) { rec_at[2 * p + e] = r; cig_at[2 * p + e] = c; r += X.nrec[2 * p + e]; c += X.ncig[2 * p + e]; }
        aln_off[p + 1] = r; cigar_off[p + 1] = c;
    }
    X.n_recs = r; X.n_cigar = c;
    if (sizes_only) return;
    X.d_rec_at.ensure_slack(n_mates); X.d_rec_at.upload(rec_at.data(), n_mates, s);
    X.d_cig_at.ensure_slack(n_mates); X.d_cig_at.upload(cig_at.data(), n_mates, s);
    X.d_pair_cig.ensure_slack(n); X.d_pair_cig.upload(pair_cig.data(), n, s);
    X.d_recs.ensure_slack(std::max<uint64_t>(r, 1)); X.d_cigar.ensure_slack(std::max<uint64_t>(c, 1));
    X.d_ob2.ensure_slack(std::max<uint64_t>(nb / 16, 1)); X.d_onm.ensure_slack(std::max<uint64_t>(nb / 32, 1));
    X.d_ob2.zero(s); X.d_onm.zero(s);
    V.rec_at = X.d_rec_at.p; V.cig_at = X.d_cig_at.p; V.pair_cig = X.d_pair_cig.p; V.recs = X.d_recs.p; V.cigar = X.d_cigar.p;
    V.out_bases2 = X.d_ob2.p; V.out_nmask = X.d_onm.p;
    ctx->timed(LCTY_K_MAP, [&] { hipLaunchKernelGGL(map_emit_kernel<true>, dim3(n_wg), dim3(64), 0, s, V); }, s);
    LCTY_HIP(hipGetLastError());
    LCTY_HIP(hipStreamSynchronize(s));                                          // rec_at & co. are host vectors of this frame
}

}  // namespace

int32_t lcty_map_reads(lcty_locus* locus, const lcty_reads_host* chunk, const lcty_map_params* params, uint64_t* aln_off, lcty_aln_rec* recs,
                       uint64_t cap_recs, uint64_t* cigar_off, uint32_t* cigar, uint64_t cap_cigar, uint32_t* bases2_out, uint32_t* nmask_out) {
    return guarded([&] {
        MapRun X;
        const bool sizes_only = !recs || !cigar || !bases2_out || !nmask_out;
        run_map(locus, chunk, params, aln_off, cigar_off, sizes_only, X);
        if (sizes_only || !chunk->n_pairs) return;
        if (X.n_recs > cap_recs || X.n_cigar > cap_cigar)
            fail(LCTY_ERR_INVALID_INPUT, "room for %llu records and %llu CIGAR words is needed", (unsigned long long)X.n_recs, (unsigned long long)X.n_cigar);
        hipStream_t s = locus->ctx->stream;
        const uint64_t nb = chunk->mate_off[2 * chunk->n_pairs];
        X.d_recs.download(recs, X.n_recs, s); X.d_cigar.download(cigar, X.n_cigar, s);
        X.d_ob2.download(bases2_out, nb / 16, s); X.d_onm.download(nmask_out, nb / 32, s);
        LCTY_HIP(hipStreamSynchronize(s));
    });
}

// the same, with the records going straight into a batch of the locus: device to device, nothing but the offsets visits the host
int32_t lcty_reads_map_append(lcty_reads* reads, const lcty_reads_host* chunk, const lcty_map_params* params) {
    return guarded([&] {
        if (!reads || !chunk || !params) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        const uint64_t n = chunk->n_pairs;
        if (n == 0) return;
        const bool trace = reads->ctx->diag_knob("map_trace", 0) != 0;
        const auto t_in = std::chrono::steady_clock::now();
        auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_in).count(); };
        // the device buffers of the mapping (arenas of CIGAR words and chains, scratch of the kernels: tens of GB for long reads on many
        // alleles) stay with the context from chunk to chunk — a streaming loop does not allocate and release them per chunk — until the
        // solver stages take the memory back (lcty_ctx::release_transfer_scratch) or lcty_ctx_trim is called
        std::lock_guard<std::mutex> one_at_a_time(reads->ctx->map_mutex);
        std::shared_ptr<MapRun> held;
        {
            std::lock_guard<std::mutex> lock(reads->ctx->scratch_mutex);
            held = std::static_pointer_cast<MapRun>(reads->ctx->map_scratch);
            if (!held) { held = std::make_shared<MapRun>(); reads->ctx->map_scratch = held; }
        }
        MapRun& X = *held;
        std::vector<uint64_t> aln_off(n + 1), cigar_off(n + 1);
        run_map(reads->locus, chunk, params, aln_off.data(), cigar_off.data(), false, X);
        if (trace) fprintf(stderr, "[lcty map] mapped at %.3f ms of the call\n", since());
        lcty_reads_host h = *chunk;
        h.aln_off = aln_off.data(); h.cigar_off = cigar_off.data(); h.recs = nullptr; h.cigar = nullptr;
        DeviceRecords dev{X.d_recs.p, X.d_cigar.p, X.d_ob2.p, X.d_onm.p, X.nrec.data(), X.max_rec_cigar};
        const auto t0 = std::chrono::steady_clock::now();
        const int32_t rc = reads_append_device(reads, &h, &dev);
        if (reads->ctx->diag_knob("map_trace", 0))
            fprintf(stderr, "[lcty map] append of the mapped chunk: %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        if (rc != LCTY_OK) fail(rc, "%s", lcty_last_error());
    });
}

}  // extern "C"
