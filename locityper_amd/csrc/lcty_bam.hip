// lcty_bam.hip — write_bam (src/model/bam.rs:356-413): the read alignments of one genotype as a coordinate-sorted BAM file + its
// BAI index. Host code: the per-read posteriors come from lcty_assignment_counts, everything a BAM record
// needs besides (names, sequences, qualities, CIGARs) from the caller's own copy of the alignment table.
//   used reads   (AllAlignments::reads, status GOOD): generate_paired_end_records / generate_single_end_records (bam.rs:233-266,
//                300-326): the locations of the read on the genotype (extend_read_gt_alns, windows.rs:762-797) folded by
//                (alignment of mate 1, alignment of mate 2) with their assignment counts (count_alignments, 144-176); one record
//                (pair) per fold: MAPQ / pr from the counts (count_to_prob, 56-67), the fold with the most counts primary
//   unused reads (AllAlignments::unused_reads, status FEW_KMERS): their PairAlignments on the genotype's contigs, us:F (268-298, 328-353)
//   tags         NM (edit distance), il (log10 likelihood of the single alignment when it differs from the location's), al (log10
//                likelihood of the location), uk (unique k-mers of the mate), pr (probability), us (used: T / F)   (bam.rs:123-141)
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "lcty_objects.hpp"

using namespace lcty;

namespace {

constexpr uint32_t UNM = 0xFFFFFFFFu;

// BGZF: gzip members of at most 64 KB with the BC extra field; virtual offset = start of the block << 16 | offset inside it
struct BgzfOut {
    FILE* f = nullptr;
    std::vector<uint8_t> block;
    uint64_t coff = 0;
    explicit BgzfOut(const char* path) {
        f = fopen(path, "wb");
        if (!f) fail(LCTY_ERR_INVALID_INPUT, "cannot create %s", path);
        block.reserve(0xff00);
    }
    ~BgzfOut() { if (f) fclose(f); }
    uint64_t tell() const { return (coff << 16) | block.size(); }
    void flush() {
        uint8_t out[0x10000];
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) fail(LCTY_ERR_RUNTIME, "zlib: deflateInit2");
        zs.next_in = block.data(); zs.avail_in = static_cast<uInt>(block.size());
        zs.next_out = out + 18; zs.avail_out = sizeof(out) - 26;
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { deflateEnd(&zs); fail(LCTY_ERR_RUNTIME, "zlib: deflate"); }
        const uint32_t clen = static_cast<uint32_t>(zs.total_out);
        deflateEnd(&zs);
        const uint32_t bsize = clen + 25;                                   // total block size - 1
        const uint8_t head[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0,
                                  static_cast<uint8_t>(bsize & 255), static_cast<uint8_t>(bsize >> 8)};
        memcpy(out, head, 18);
        const uint32_t crc = static_cast<uint32_t>(crc32(crc32(0L, Z_NULL, 0), block.data(), static_cast<uInt>(block.size())));
        const uint32_t isize = static_cast<uint32_t>(block.size());
        uint8_t* tail = out + 18 + clen;
        for (int i = 0; i < 4; i++) { tail[i] = static_cast<uint8_t>(crc >> (8 * i)); tail[4 + i] = static_cast<uint8_t>(isize >> (8 * i)); }
        if (fwrite(out, 1, clen + 26, f) != clen + 26) fail(LCTY_ERR_RUNTIME, "write error");
        coff += clen + 26;
        block.clear();
    }
    void write(const void* p, size_t n) {
        const uint8_t* b = static_cast<const uint8_t*>(p);
        while (n) {
            const size_t take = std::min<size_t>(n, 0xff00 - block.size());
            block.insert(block.end(), b, b + take);
            b += take; n -= take;
            if (block.size() == 0xff00) flush();
        }
    }
    void close() {
        if (!block.empty()) flush();
        flush();                                                            // the empty block that marks the end of a BGZF file
        if (fclose(f) != 0) { f = nullptr; fail(LCTY_ERR_RUNTIME, "write error"); }
        f = nullptr;
    }
};

template <typename T> void put(std::vector<uint8_t>& v, T x) { const uint8_t* p = reinterpret_cast<const uint8_t*>(&x); v.insert(v.end(), p, p + sizeof(T)); }

// reg2bin of the SAM specification (section 5.3)
uint32_t reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return static_cast<uint32_t>(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return static_cast<uint32_t>(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return static_cast<uint32_t>(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return static_cast<uint32_t>(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return static_cast<uint32_t>(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

struct Rec {                    // one BAM record before serialisation
    int32_t tid = -1, pos = 0, mtid = -1, mpos = 0, tlen = 0;
    uint32_t flag = 0, mapq = 0, end = 0;
    uint64_t pair = 0; uint32_t mate = 0;        // whose name / sequence
    bool opposite = false;                        // sequence on the other strand than stored
    const uint32_t* cigar = nullptr; uint32_t n_cigar = 0;
    std::vector<uint8_t> aux;
    uint64_t order = 0;
};

struct AlnInfo { uint32_t edit, ref_len; double ln_prob; bool ok; };

// count_region_operations_fast + limited_clipping + edit_distance + ErrorProfile::ln_prob of one record (aln.rs:288-317,
// err_prof.rs:73-79, 212-221): the same arithmetic as the scoring kernel, on the host
AlnInfo score_record(const lcty_bg& bg, const lcty_aln_rec& rc, const uint32_t* cg, uint32_t contig_len) {
    uint32_t matches = 0, mism = 0, ins = 0, del = 0, left = 0, right = 0;
    const uint32_t n = rc.n_cigar;
    bool ok = n > 0;
    for (uint32_t t = 0; t < n; t++) {
        uint32_t op = cg[t] & 15u, len = cg[t] >> 4;
        if (op == LCTY_CIGAR_H && (t == 0 || t + 1 == n)) op = LCTY_CIGAR_S;
        switch (op) {
            case LCTY_CIGAR_EQ: matches += len; break;
            case LCTY_CIGAR_X: mism += len; break;
            case LCTY_CIGAR_I: ins += len; break;
            case LCTY_CIGAR_D: del += len; break;
            case LCTY_CIGAR_S: if (t == 0) left = len; else if (t + 1 == n) right = len; break;
            default: ok = false;
        }
    }
    const uint32_t ref_len = matches + mism + del, end = rc.pos + ref_len;
    const uint32_t clip = std::min(left, rc.pos) + std::min(right, contig_len > end ? contig_len - end : 0u);
    AlnInfo a;
    a.edit = mism + ins + clip + del; a.ref_len = ref_len; a.ok = ok;
    a.ln_prob = bg.op_lnprobs[0] * static_cast<double>(matches) + bg.op_lnprobs[1] * static_cast<double>(mism)
              + bg.op_lnprobs[2] * static_cast<double>(ins) + bg.op_lnprobs[3] * static_cast<double>(del)
              + bg.op_lnprobs[4] * static_cast<double>(clip);
    return a;
}

}  // namespace

extern "C" {

int32_t lcty_write_bam(const char* path, lcty_reads* reads, const lcty_reads_host* table, const uint64_t* name_off, const char* names,
                       const uint64_t* qual_off, const uint8_t* quals, const char* const* allele_names, const uint16_t* genotype,
                       uint32_t ploidy, uint16_t attempts, const uint64_t* read_off, const uint16_t* counts, uint64_t* n_records) {
    return guarded([&] {
        if (!path || !reads || !table || !name_off || !names || !allele_names || !genotype || !read_off || !counts || ploidy == 0)
            fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (!reads->scored) fail(LCTY_ERR_INVALID_INPUT, "lcty_score_reads has not been called on this batch");
        if (reads->counted) fail(LCTY_ERR_UNSUPPORTED, "a batch of counted alignments has no CIGARs to write");
        if (table->n_pairs != reads->n_pairs) fail(LCTY_ERR_INVALID_INPUT, "the table has %llu read pairs, the batch %llu",
                                                    static_cast<unsigned long long>(table->n_pairs), static_cast<unsigned long long>(reads->n_pairs));
        if (table->aln_off[table->n_pairs] != reads->n_recs)
            fail(LCTY_ERR_UNSUPPORTED, "the batch holds %llu records, the table %llu: after lcty_recover_alignments the transferred alignments are "
                 "not in the caller's table", static_cast<unsigned long long>(reads->n_recs), static_cast<unsigned long long>(table->aln_off[table->n_pairs]));
        lcty_locus* loc = reads->locus;
        const lcty_params& prm = loc->prm;
        const bool paired = loc->bg.is_paired != 0;
        const uint32_t A = loc->n_alleles;
        const uint64_t R = reads->n_pairs;
        // create_bam_header (bam.rs:24-51): the contigs of the genotype once each, in genotype order
        std::vector<int32_t> contig_to_tid(A, -1);
        std::vector<uint32_t> unique_ids;
        std::string text;
        for (uint32_t t = 0; t < ploidy; t++) {
            const uint32_t id = genotype[t];
            if (id >= A) fail(LCTY_ERR_INVALID_INPUT, "genotype refers to allele %u >= %u", id, A);
            if (contig_to_tid[id] >= 0) continue;
            contig_to_tid[id] = static_cast<int32_t>(unique_ids.size());
            unique_ids.push_back(id);
            text += std::string("@SQ\tSN:") + allele_names[id] + "\tLN:" + std::to_string(loc->allele_len[id]) + "\n";
        }
        // products of the batch
        std::vector<uint8_t> status(R); std::vector<double> unmapped(R); std::vector<uint16_t> uniq(2 * R);
        std::vector<uint64_t> pa_off(R + 1);
        auto ok = [](int32_t rc) { if (rc != LCTY_OK) throw Error(rc, std::string(lcty_last_error())); };
        ok(lcty_reads_get_status(reads, status.data(), nullptr, unmapped.data(), uniq.data()));
        ok(lcty_reads_get_pair_alns(reads, pa_off.data(), nullptr, 0));
        std::vector<lcty_pair_aln> pa(std::max<uint64_t>(pa_off[R], 1));
        ok(lcty_reads_get_pair_alns(reads, pa_off.data(), pa.data(), pa_off[R]));

        std::vector<Rec> recs;
        const double INV_LN10 = 1.0 / std::log(10.0);
        uint64_t order = 0;
        // create_record (bam.rs:89-132) for one mate
        auto make = [&](uint64_t r, uint32_t mate, uint32_t ix, double total_prob, const std::vector<AlnInfo>& info) {
            Rec rec;
            rec.pair = r; rec.mate = mate; rec.order = order++;
            const uint64_t a0 = table->aln_off[r];
            // MateData::strand = strand of the mate's primary record; the sequence is stored as in that record
            uint64_t prim = a0;
            if (mate == 1)
                for (uint64_t q = a0 + 1; q < table->aln_off[r + 1]; q++)
                    if (!(table->recs[q].flags & (LCTY_FLAG_SECONDARY | LCTY_FLAG_SUPPL))) { prim = q; break; }
            const bool stored_rev = (table->recs[prim].flags & LCTY_FLAG_REVERSE) != 0;
            if (ix != UNM) {
                const lcty_aln_rec& rc = table->recs[a0 + ix];
                rec.tid = contig_to_tid[rc.contig];
                if (rec.tid < 0) fail(LCTY_ERR_RUNTIME, "Contig ID undefined");
                rec.pos = static_cast<int32_t>(rc.pos);
                const bool rev = (rc.flags & LCTY_FLAG_REVERSE) != 0;
                if (rev) rec.flag |= 0x10;
                rec.opposite = rev != stored_rev;
                rec.cigar = table->cigar + table->cigar_off[r] + rc.cigar_rel; rec.n_cigar = rc.n_cigar;
                rec.end = rc.pos + info[ix].ref_len;
                rec.aux.insert(rec.aux.end(), {'N', 'M', 'I'}); put<uint32_t>(rec.aux, info[ix].edit);
                if (info[ix].ln_prob != total_prob) { rec.aux.insert(rec.aux.end(), {'i', 'l', 'f'}); put<float>(rec.aux, static_cast<float>(info[ix].ln_prob * INV_LN10)); }
            } else {
                rec.flag |= 0x4; rec.tid = -1; rec.pos = 0;
                rec.opposite = stored_rev;                         // Strand::Forward
            }
            rec.aux.insert(rec.aux.end(), {'a', 'l', 'f'}); put<float>(rec.aux, static_cast<float>(total_prob * INV_LN10));
            rec.aux.insert(rec.aux.end(), {'u', 'k', 'S'}); put<uint16_t>(rec.aux, uniq[2 * r + mate]);
            return rec;
        };
        // connect_pair (bam.rs:178-221)
        auto connect = [&](Rec& r1, Rec& r2, int64_t insert) {
            const bool m1 = !(r1.flag & 0x4), m2 = !(r2.flag & 0x4);
            if (!m1 && !m2) { r1.flag |= 0x8; r2.flag |= 0x8; }
            else if (!m1) { r2.flag |= 0x8; r1.tid = r2.tid; r1.pos = r2.pos; }
            else if (!m2) { r1.flag |= 0x8; r2.tid = r1.tid; r2.pos = r1.pos; }
            else {
                r1.flag |= 0x2; r2.flag |= 0x2;
                if (r1.flag & 0x10) r2.flag |= 0x20;
                if (r2.flag & 0x10) r1.flag |= 0x20;
            }
            r1.flag |= 0x1 | 0x40; r1.mtid = r2.tid; r1.mpos = r2.pos; r1.tlen = static_cast<int32_t>(insert);
            r2.flag |= 0x1 | 0x80; r2.mtid = r1.tid; r2.mpos = r1.pos; r2.tlen = static_cast<int32_t>(-insert);
        };
        auto insert_size = [&](uint64_t r, uint32_t i1, uint32_t i2, const std::vector<AlnInfo>& info) -> int64_t {   // calc_insert_size (69-86)
            if (i1 == UNM || i2 == UNM) return 0;
            const lcty_aln_rec& a = table->recs[table->aln_off[r] + i1]; const lcty_aln_rec& b = table->recs[table->aln_off[r] + i2];
            const int64_t s1 = a.pos, e1 = a.pos + info[i1].ref_len, s2 = b.pos, e2 = b.pos + info[i2].ref_len;
            return s1 <= s2 ? e2 - s1 : s2 - e1;
        };

        uint64_t g = 0;                                             // index among the used reads
        // bam.rs:372-398: the records of ALL used reads first, then those of the unused ones — the stable sort below keeps that order
        // among records of one position
        for (int pass = 0; pass < 2; pass++)
        for (uint64_t r = 0; r < R; r++) {
            const bool used = status[r] == LCTY_READ_GOOD, unused = status[r] == LCTY_READ_FEW_KMERS;
            if (!used && !unused) continue;
            if ((pass == 0) != used) continue;
            // the single alignments of the pair, normalised per read end (normalize_probs, locs.rs:358-360)
            const uint64_t a0 = table->aln_off[r], a1 = table->aln_off[r + 1];
            std::vector<AlnInfo> info(a1 - a0);
            double best[2] = {-INFINITY, -INFINITY};
            uint32_t second = static_cast<uint32_t>(a1 - a0);
            for (uint64_t q = a0 + 1; q < a1; q++)
                if (!(table->recs[q].flags & (LCTY_FLAG_SECONDARY | LCTY_FLAG_SUPPL))) { second = static_cast<uint32_t>(q - a0); break; }
            for (uint64_t q = a0; q < a1; q++) {
                const lcty_aln_rec& rc = table->recs[q];
                if (rc.contig >= A) { info[q - a0] = AlnInfo{0, 0, 0.0, false}; continue; }
                info[q - a0] = score_record(loc->bg, rc, table->cigar + table->cigar_off[r] + rc.cigar_rel, loc->allele_len[rc.contig]);
                const int e = (q - a0) >= second ? 1 : 0;
                if (info[q - a0].ok) best[e] = std::fmax(best[e], info[q - a0].ln_prob);
            }
            for (uint64_t q = a0; q < a1; q++) info[q - a0].ln_prob -= best[(q - a0) >= second ? 1 : 0];
            const lcty_pair_aln* seg = pa.data() + pa_off[r];
            const uint64_t nseg = pa_off[r + 1] - pa_off[r];
            auto contig_range = [&](uint32_t id, uint64_t* lo, uint64_t* hi) {
                uint64_t x = 0;
                while (x < nseg && seg[x].contig < id) x++;
                *lo = x;
                while (x < nseg && seg[x].contig == id) x++;
                *hi = x;
            };
            if (used) {
                // extend_read_gt_alns (windows.rs:762-797): the read's locations on the genotype, then count_alignments (bam.rs:144-176)
                struct LocItem { double lp; uint32_t i1, i2; uint32_t push; };
                std::vector<LocItem> locs;
                const double unm = unmapped[r];
                double thresh = unm - prm.prob_diff;
                for (uint32_t t = 0; t < ploidy; t++) {
                    uint64_t lo, hi;
                    contig_range(genotype[t], &lo, &hi);
                    if (lo < hi) {
                        thresh = std::fmax(thresh, seg[lo].ln_prob - prm.prob_diff);
                        for (uint64_t x = lo; x < hi && seg[x].ln_prob >= thresh; x++)
                            locs.push_back(LocItem{seg[x].ln_prob, seg[x].ix1, paired ? seg[x].ix2 : UNM, static_cast<uint32_t>(locs.size())});
                    }
                }
                if (unm >= thresh) locs.push_back(LocItem{unm, UNM, UNM, static_cast<uint32_t>(locs.size())});
                std::stable_sort(locs.begin(), locs.end(), [](const LocItem& a, const LocItem& b) { return a.lp > b.lp; });
                while (!locs.empty() && !(locs.back().lp >= thresh)) locs.pop_back();
                const uint64_t c0 = read_off[g], c1 = read_off[g + 1];
                if (c1 - c0 != locs.size())
                    fail(LCTY_ERR_INVALID_INPUT, "read pair %llu has %zu locations on the genotype, the counts give %llu: counts of another genotype?",
                         static_cast<unsigned long long>(r), locs.size(), static_cast<unsigned long long>(c1 - c0));
                std::map<std::pair<uint32_t, uint32_t>, std::pair<uint32_t, double>> fold;
                std::pair<uint32_t, uint32_t> best_ij{UNM, UNM};
                std::pair<uint32_t, double> best_val{0, -INFINITY};
                for (size_t t = 0; t < locs.size(); t++) {
                    const uint32_t count = counts[c0 + t];
                    if (count == 0) continue;
                    const std::pair<uint32_t, uint32_t> ij{locs[t].i1, locs[t].i2};
                    auto it = fold.find(ij);
                    if (it == fold.end()) it = fold.emplace(ij, std::make_pair(count, locs[t].lp)).first;
                    else it->second.first += count;
                    if (it->second > best_val) { best_ij = ij; best_val = it->second; }
                }
                for (const auto& kv : fold) {
                    const uint32_t i1 = kv.first.first, i2 = kv.first.second, count = kv.second.first;
                    const double lp = kv.second.second;
                    // count_to_prob (bam.rs:56-67)
                    float prob; uint32_t mapq;
                    if (count == attempts) { prob = 1.0f; mapq = 60; }
                    else {
                        if (count > attempts) fail(LCTY_ERR_INVALID_INPUT, "count %u of %u attempts", count, attempts);
                        prob = static_cast<float>(count) / static_cast<float>(attempts);
                        mapq = static_cast<uint32_t>(std::fmin(std::round(-10.0f * std::log10(1.0f - prob)), 60.0f));
                    }
                    Rec r1 = make(r, 0, i1, lp, info);
                    auto finish = [&](Rec& x) {
                        x.mapq = mapq;
                        x.aux.insert(x.aux.end(), {'p', 'r', 'f'}); put<float>(x.aux, prob);
                        x.aux.insert(x.aux.end(), {'u', 's', 'A', 'T'});
                        if (kv.first != best_ij) x.flag |= 0x100;
                    };
                    finish(r1);
                    if (paired) {
                        Rec r2 = make(r, 1, i2, lp, info);
                        finish(r2);
                        connect(r1, r2, insert_size(r, i1, i2, info));
                        recs.push_back(std::move(r1)); recs.push_back(std::move(r2));
                    } else recs.push_back(std::move(r1));
                }
                g++;
            } else {
                // generate_unused_*_records (bam.rs:268-298, 328-353): the PairAlignments on the genotype's contigs, best first
                std::vector<const lcty_pair_aln*> alns;
                for (const uint32_t id : unique_ids) {
                    uint64_t lo, hi;
                    contig_range(id, &lo, &hi);
                    for (uint64_t x = lo; x < hi; x++) alns.push_back(&seg[x]);
                }
                std::stable_sort(alns.begin(), alns.end(), [](const lcty_pair_aln* a, const lcty_pair_aln* b) { return a->ln_prob > b->ln_prob; });
                bool secondary = false;
                for (const lcty_pair_aln* x : alns) {
                    Rec r1 = make(r, 0, x->ix1, x->ln_prob, info);
                    auto finish = [&](Rec& y) { if (secondary) y.flag |= 0x100; y.aux.insert(y.aux.end(), {'u', 's', 'A', 'F'}); };
                    if (paired) {
                        Rec r2 = make(r, 1, x->ix2, x->ln_prob, info);
                        if (secondary) { r1.flag |= 0x100; r2.flag |= 0x100; }
                        connect(r1, r2, insert_size(r, x->ix1, x->ix2, info));
                        r1.aux.insert(r1.aux.end(), {'u', 's', 'A', 'F'}); r2.aux.insert(r2.aux.end(), {'u', 's', 'A', 'F'});
                        recs.push_back(std::move(r1)); recs.push_back(std::move(r2));
                    } else { finish(r1); recs.push_back(std::move(r1)); }
                    secondary = true;
                }
            }
        }
        // stable sort by (tid as u32, pos): mates stay together, unmapped pairs at the end (bam.rs:401-403)
        std::stable_sort(recs.begin(), recs.end(), [](const Rec& a, const Rec& b) {
            const uint32_t ta = static_cast<uint32_t>(a.tid), tb = static_cast<uint32_t>(b.tid);
            return ta != tb ? ta < tb : a.pos < b.pos;
        });

        // ---- the file
        BgzfOut out(path);
        std::vector<uint8_t> buf;
        buf.insert(buf.end(), {'B', 'A', 'M', 1});
        put<uint32_t>(buf, static_cast<uint32_t>(text.size())); buf.insert(buf.end(), text.begin(), text.end());
        put<uint32_t>(buf, static_cast<uint32_t>(unique_ids.size()));
        for (const uint32_t id : unique_ids) {
            const std::string nm = allele_names[id];
            put<uint32_t>(buf, static_cast<uint32_t>(nm.size() + 1)); buf.insert(buf.end(), nm.begin(), nm.end()); buf.push_back(0);
            put<uint32_t>(buf, loc->allele_len[id]);
        }
        out.write(buf.data(), buf.size());
        // BAI: per reference bins -> chunks, 16-kb linear index
        struct RefIndex { std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins; std::vector<uint64_t> linear; uint64_t n_mapped = 0, n_unmapped = 0, off_beg = 0, off_end = 0; };
        std::vector<RefIndex> index(unique_ids.size());
        uint64_t n_no_coor = 0;
        auto comp = [](char c) -> char {
            switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; }
        };
        static const uint8_t CODE_A = 1, CODE_C = 2, CODE_G = 4, CODE_T = 8, CODE_N = 15;
        for (const Rec& rc : recs) {
            const uint64_t m = 2 * rc.pair + rc.mate;
            const uint32_t len = table->mate_len[m];
            const uint64_t off = table->mate_off[m];
            const std::string name(names + name_off[rc.pair], names + name_off[rc.pair + 1]);
            std::string seq(len, 'N');
            for (uint32_t i = 0; i < len; i++) {
                const uint64_t b = off + i;
                const bool other = (table->nmask[b >> 5] >> (b & 31)) & 1u;
                seq[i] = other ? 'N' : "ACGT"[(table->bases2[b >> 4] >> (2 * (b & 15))) & 3u];
            }
            std::vector<uint8_t> q(len, 255);
            if (quals && qual_off) for (uint32_t i = 0; i < len && qual_off[m] + i < qual_off[m + 1]; i++) q[i] = quals[qual_off[m] + i];
            if (rc.opposite) {                                          // MateData::get_seq_and_qual on the other strand (locs.rs:88-103)
                std::reverse(seq.begin(), seq.end());
                for (auto& c : seq) c = comp(c);
                std::reverse(q.begin(), q.end());
            }
            const bool mapped = !(rc.flag & 0x4);
            const int64_t beg = rc.pos, end = mapped ? std::max<int64_t>(rc.end, beg + 1) : beg + 1;
            buf.clear();
            put<int32_t>(buf, rc.tid); put<int32_t>(buf, rc.pos);
            buf.push_back(static_cast<uint8_t>(name.size() + 1)); buf.push_back(static_cast<uint8_t>(rc.mapq));
            put<uint16_t>(buf, static_cast<uint16_t>(rc.tid >= 0 ? reg2bin(beg, end) : 4680));
            put<uint16_t>(buf, static_cast<uint16_t>(mapped ? rc.n_cigar : 0)); put<uint16_t>(buf, static_cast<uint16_t>(rc.flag));
            put<uint32_t>(buf, len); put<int32_t>(buf, rc.mtid); put<int32_t>(buf, rc.mpos); put<int32_t>(buf, rc.tlen);
            buf.insert(buf.end(), name.begin(), name.end()); buf.push_back(0);
            if (mapped) for (uint32_t i = 0; i < rc.n_cigar; i++) put<uint32_t>(buf, rc.cigar[i]);
            for (uint32_t i = 0; i < len; i += 2) {
                auto code = [&](char c) -> uint8_t { return c == 'A' ? CODE_A : c == 'C' ? CODE_C : c == 'G' ? CODE_G : c == 'T' ? CODE_T : CODE_N; };
                buf.push_back(static_cast<uint8_t>((code(seq[i]) << 4) | (i + 1 < len ? code(seq[i + 1]) : 0)));
            }
            buf.insert(buf.end(), q.begin(), q.end());
            buf.insert(buf.end(), rc.aux.begin(), rc.aux.end());
            if (name.size() + 1 > 255 || rc.n_cigar > 65535) fail(LCTY_ERR_UNSUPPORTED, "read name or CIGAR too long for a BAM record");
            const uint64_t v0 = out.tell();
            const uint32_t bs = static_cast<uint32_t>(buf.size());
            out.write(&bs, 4); out.write(buf.data(), buf.size());
            const uint64_t v1 = out.tell();
            if (rc.tid >= 0) {
                RefIndex& ri = index[static_cast<size_t>(rc.tid)];
                auto& chunks = ri.bins[reg2bin(beg, end)];
                if (!chunks.empty() && chunks.back().second == v0) chunks.back().second = v1; else chunks.emplace_back(v0, v1);
                for (int64_t w = beg >> 14; w <= (end - 1) >> 14; w++) {
                    if (ri.linear.size() <= static_cast<size_t>(w)) ri.linear.resize(static_cast<size_t>(w) + 1, 0);
                    if (ri.linear[static_cast<size_t>(w)] == 0) ri.linear[static_cast<size_t>(w)] = v0;
                }
                if (ri.n_mapped + ri.n_unmapped == 0) ri.off_beg = v0;
                ri.off_end = v1;
                if (mapped) ri.n_mapped++; else ri.n_unmapped++;
            } else n_no_coor++;
        }
        out.close();
        // bam::index::build(.., Bai, ..) (bam.rs:410): magic, per reference {bins (+ the 37450 pseudo-bin), linear index}, n_no_coor
        {
            std::vector<uint8_t> bai;
            bai.insert(bai.end(), {'B', 'A', 'I', 1});
            put<uint32_t>(bai, static_cast<uint32_t>(index.size()));
            for (RefIndex& ri : index) {
                const bool any = ri.n_mapped + ri.n_unmapped > 0;
                put<uint32_t>(bai, static_cast<uint32_t>(ri.bins.size() + (any ? 1 : 0)));
                for (const auto& b : ri.bins) {
                    put<uint32_t>(bai, b.first); put<uint32_t>(bai, static_cast<uint32_t>(b.second.size()));
                    for (const auto& c : b.second) { put<uint64_t>(bai, c.first); put<uint64_t>(bai, c.second); }
                }
                if (any) {
                    put<uint32_t>(bai, 37450u); put<uint32_t>(bai, 2u);
                    put<uint64_t>(bai, ri.off_beg); put<uint64_t>(bai, ri.off_end); put<uint64_t>(bai, ri.n_mapped); put<uint64_t>(bai, ri.n_unmapped);
                }
                for (size_t w = 1; w < ri.linear.size(); w++) if (ri.linear[w] == 0) ri.linear[w] = ri.linear[w - 1];
                put<uint32_t>(bai, static_cast<uint32_t>(ri.linear.size()));
                for (const uint64_t v : ri.linear) put<uint64_t>(bai, v);
            }
            put<uint64_t>(bai, n_no_coor);
            const std::string bai_path = std::string(path) + ".bai";
            FILE* f = fopen(bai_path.c_str(), "wb");
            if (!f) fail(LCTY_ERR_INVALID_INPUT, "cannot create %s", bai_path.c_str());
            const bool bad = fwrite(bai.data(), 1, bai.size(), f) != bai.size();
            if (fclose(f) != 0 || bad) fail(LCTY_ERR_RUNTIME, "write error on %s", bai_path.c_str());
        }
        if (n_records) *n_records = recs.size();
    });
}

}  // extern "C"
