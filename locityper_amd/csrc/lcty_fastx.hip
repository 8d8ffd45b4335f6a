// lcty_fastx.hip — the readers and writers AROUND read recruitment (host code, no device):
//   src/seq/fastx.rs:232-470   Reader::read_next (FASTA / FASTQ, name = the header up to the first blank), PairedEndInterleaved,
//                              PairedEndReaders with their checks (odd number of records, names that do not match, files of
//                              different length, a record without '+', sequence and qualities of different length)
//   src/seq/fastx.rs:58-75, 141-150, 262-272   write_fasta / write_fastq, `[T; 2]::write_to`: the two mates one after the other
//   src/seq/recruit.rs:1000-1030   recruit_single_thread's loop: read -> recruit -> `record.write_to(writers.get(locus_ix))`
// The records of a chunk come out packed the way lcty_recruit takes them (2 bits per base, 32-base aligned mates, one "not ACGT" bit per
// base: kmers() treats every byte but the four capitals as N, seq/kmers.rs:178-190); their text stays with the handle until the next
// chunk so that the recruited ones can be written, byte for byte as they were read (the header's description is dropped, as upstream).
// Containers: plain and gzip (zlib reads both); the reference's .lz4 / .br inputs are refused with LCTY_ERR_UNSUPPORTED.
#include <zlib.h>

#include <memory>
#include <string>
#include <vector>

#include "lcty_common.hpp"

using namespace lcty;

namespace {

struct LineReader {
    gzFile f = nullptr;
    std::string path;
    std::vector<char> buf;
    size_t at = 0, end = 0;
    bool eof = false;
    ~LineReader() { if (f) gzclose(f); }
    void open(const char* p) {
        path = p;
        const size_t n = path.size();
        if ((n > 4 && path.compare(n - 4, 4, ".lz4") == 0) || (n > 3 && path.compare(n - 3, 3, ".br") == 0))
            fail(LCTY_ERR_UNSUPPORTED, "%s: FASTA / FASTQ input is read plain or gzip-compressed", p);
        f = gzopen(p, "rb");
        if (!f) fail(LCTY_ERR_INVALID_INPUT, "cannot open %s", p);
        gzbuffer(f, 1u << 20);
        buf.resize(1u << 20);
    }
    // read_line (fastx.rs:79-99): the line without its end ("\n" or "\r\n"); false at the end of the file with nothing read
    bool line(std::string& out) {
        out.clear();
        bool any = false;
        for (;;) {
            if (at == end) {
                if (eof) break;
                const int n = gzread(f, buf.data(), static_cast<unsigned>(buf.size()));
                if (n < 0) { int e = 0; fail(LCTY_ERR_INVALID_DATA, "%s: %s", path.c_str(), gzerror(f, &e)); }
                if (n == 0) { eof = true; break; }
                at = 0; end = static_cast<size_t>(n);
            }
            const char* nl = static_cast<const char*>(memchr(buf.data() + at, '\n', end - at));
            if (nl) {
                out.append(buf.data() + at, static_cast<size_t>(nl - (buf.data() + at)));
                at = static_cast<size_t>(nl - buf.data()) + 1;
                if (!out.empty() && out.back() == '\r') out.pop_back();
                return true;
            }
            out.append(buf.data() + at, end - at);
            any = true;
            at = end;
        }
        return any || !out.empty();
    }
};

struct Record { std::string name, seq, qual; bool fastq = false; };

// Reader (fastx.rs:296-440): `buffer` holds the header line of the next record
struct Reader {
    LineReader in;
    std::string header;
    bool has_header = false;
    void open(const char* p) { in.open(p); has_header = in.line(header) && !header.empty(); }
    bool next(Record& r) {
        r.name.clear(); r.seq.clear(); r.qual.clear();
        if (!has_header) return false;                                       // "If it is empty, the file has ended" (fastx.rs:413-415)
        const size_t sp = header.find(' ');
        r.name.assign(header, 1, (sp == std::string::npos ? header.size() : sp) - 1);      // only the name before any description (417-419)
        std::string l;
        if (header[0] == '>') {                                              // fill_fasta_record (321-341)
            r.fastq = false;
            has_header = false;
            while (in.line(l)) {
                if (!l.empty() && (l[0] == '>' || l[0] == '@')) { header = l; has_header = true; break; }
                r.seq += l;
            }
            if (r.seq.empty() && !has_header) fail(LCTY_ERR_INVALID_DATA, "Fasta record %s has an empty sequence.", r.name.c_str());
            return true;
        }
        r.fastq = true;                                                      // fill_fastq_record (343-368)
        in.line(r.seq);
        if (!in.line(l)) fail(LCTY_ERR_INVALID_DATA, "Fastq record %s is incomplete", r.name.c_str());
        if (l.empty() || l[0] != '+') fail(LCTY_ERR_INVALID_DATA, "Fastq record %s has incorrect format", r.name.c_str());
        in.line(r.qual);
        if (r.seq.size() != r.qual.size()) fail(LCTY_ERR_INVALID_DATA, "Fastq record %s has non-matching sequence and qualities", r.name.c_str());
        has_header = in.line(header) && !header.empty();
        return true;
    }
};

// equal_names_fast (fastx.rs:101-109): same length and, beyond three characters, the same third character from the end
bool equal_names_fast(const std::string& a, const std::string& b) {
    return a.size() == b.size() && (a.size() <= 3 || a[a.size() - 3] == b[b.size() - 3]);
}

}  // namespace

struct lcty_fastx {
    Reader r1, r2;
    bool two_files = false, interleaved = false, paired = false;
    std::string what;                                   // the file name(s) for messages
    // the current chunk: text and packed form
    std::vector<Record> recs;                           // 2 per pair when paired
    std::vector<uint32_t> mate_len; std::vector<uint64_t> mate_off, zero_off;
    std::vector<uint32_t> bases2, nmask;
    uint64_t n = 0;
};

struct lcty_fastx_writers {
    std::vector<gzFile> gz; std::vector<FILE*> plain; std::vector<std::string> paths;
    ~lcty_fastx_writers() {
        for (gzFile g : gz) if (g) gzclose(g);
        for (FILE* f : plain) if (f) fclose(f);
    }
    void put(uint32_t i, const std::string& s) {
        if (gz[i]) { if (gzwrite(gz[i], s.data(), static_cast<unsigned>(s.size())) != static_cast<int>(s.size())) fail(LCTY_ERR_RUNTIME, "write error on %s", paths[i].c_str()); }
        else if (fwrite(s.data(), 1, s.size(), plain[i]) != s.size()) fail(LCTY_ERR_RUNTIME, "write error on %s", paths[i].c_str());
    }
};

namespace {

void record_text(const Record& r, std::string& out) {                        // write_fastq / write_fasta (fastx.rs:46-75, 262-272)
    if (r.qual.empty()) { out += '>'; out += r.name; out += '\n'; out += r.seq; out += '\n'; }
    else { out += '@'; out += r.name; out += '\n'; out += r.seq; out += "\n+\n"; out += r.qual; out += '\n'; }
}

bool read_unit(lcty_fastx* f, Record& a, Record& b) {
    if (!f->paired) return f->r1.next(a);
    if (f->interleaved) {                                                    // PairedEndInterleaved::read_next (fastx.rs:444-466)
        if (!f->r1.next(a)) return false;
        if (!f->r1.next(b)) fail(LCTY_ERR_INVALID_DATA, "Odd number of records in an interleaved input file(s) %s", f->what.c_str());
        if (!equal_names_fast(a.name, b.name))
            fail(LCTY_ERR_INVALID_DATA, "Interleaved input file(s) %s contains non matching first and second mate (%s and %s)", f->what.c_str(),
                 a.name.c_str(), b.name.c_str());
        return true;
    }
    const bool g1 = f->r1.next(a), g2 = f->r2.next(b);                       // PairedEndReaders::read_next (fastx.rs:490-511)
    if (!g1 && !g2) return false;
    if (g1 != g2) fail(LCTY_ERR_INVALID_DATA, "Different number of records in paired-end input files %s", f->what.c_str());
    if (!equal_names_fast(a.name, b.name))
        fail(LCTY_ERR_INVALID_DATA, "Paired-end input files %s have non matching first and second mates (%s and %s)", f->what.c_str(), a.name.c_str(),
             b.name.c_str());
    return true;
}

void pack(lcty_fastx* f) {
    const size_t m = f->recs.size();                                          // mates
    const size_t per = f->paired ? 2 : 1;
    const uint64_t n = m / per;
    f->n = n;
    f->mate_len.assign(2 * n, 0); f->mate_off.assign(2 * n + 1, 0); f->zero_off.assign(n + 1, 0);
    uint64_t off = 0;
    for (uint64_t i = 0; i < n; i++)
        for (size_t e = 0; e < 2; e++) {
            f->mate_off[2 * i + e] = off;
            if (e < per) {
                const size_t len = f->recs[i * per + e].seq.size();
                if (len > 0xFFFFFFFFull) fail(LCTY_ERR_UNSUPPORTED, "a read of more than 2^32 bases");
                f->mate_len[2 * i + e] = static_cast<uint32_t>(len);
                off += (len + 31) / 32 * 32;
            }
        }
    f->mate_off[2 * n] = off;
    f->bases2.assign(off / 16 + 1, 0); f->nmask.assign(off / 32 + 1, 0);
    for (uint64_t i = 0; i < n; i++)
        for (size_t e = 0; e < per; e++) {
            const std::string& s = f->recs[i * per + e].seq;
            const uint64_t o = f->mate_off[2 * i + e];
            for (size_t k = 0; k < s.size(); k++) {
                uint32_t code = 0; bool other = false;
                switch (s[k]) { case 'A': code = 0; break; case 'C': code = 1; break; case 'G': code = 2; break; case 'T': code = 3; break; default: other = true; }
                const uint64_t at = o + k;
                f->bases2[at >> 4] |= code << (2 * (at & 15));
                if (other) f->nmask[at >> 5] |= 1u << (at & 31);
            }
        }
}

}  // namespace

extern "C" {

int32_t lcty_fastx_open(const char* path1, const char* path2, int32_t interleaved, lcty_fastx** out) {
    return guarded([&] {
        if (!path1 || !out) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (path2 && interleaved) fail(LCTY_ERR_INVALID_INPUT, "two input files cannot be interleaved as well");
        std::unique_ptr<lcty_fastx> f(new lcty_fastx());
        f->r1.open(path1);
        f->what = path1;
        if (path2) { f->r2.open(path2); f->two_files = true; f->what += std::string(" and ") + path2; }
        f->interleaved = interleaved != 0;
        f->paired = f->two_files || f->interleaved;
        *out = f.release();
    });
}

void lcty_fastx_close(lcty_fastx* f) { delete f; }

int32_t lcty_fastx_next(lcty_fastx* f, uint64_t max_records, lcty_reads_host* view, uint64_t* n) {
    return guarded([&] {
        if (!f || !view || !n) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        f->recs.clear();
        Record a, b;
        for (uint64_t i = 0; i < max_records && read_unit(f, a, b); i++) {
            f->recs.push_back(a);
            if (f->paired) f->recs.push_back(b);
        }
        pack(f);
        memset(view, 0, sizeof(*view));
        view->n_pairs = f->n;
        view->mate_len = f->mate_len.data(); view->mate_off = f->mate_off.data(); view->bases2 = f->bases2.data(); view->nmask = f->nmask.data();
        view->aln_off = f->zero_off.data(); view->cigar_off = f->zero_off.data();
        *n = f->n;
    });
}

int32_t lcty_fastx_is_paired(const lcty_fastx* f, int32_t* paired) {
    return guarded([&] {
        if (!f || !paired) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        *paired = f->paired ? 1 : 0;
    });
}

int32_t lcty_fastx_writers_open(const char* const* paths, uint32_t n, lcty_fastx_writers** out) {
    return guarded([&] {
        if (!paths || !out) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        std::unique_ptr<lcty_fastx_writers> w(new lcty_fastx_writers());
        w->gz.assign(n, nullptr); w->plain.assign(n, nullptr); w->paths.resize(n);
        for (uint32_t i = 0; i < n; i++) {
            if (!paths[i]) fail(LCTY_ERR_INVALID_INPUT, "null argument");
            w->paths[i] = paths[i];
            const size_t len = w->paths[i].size();
            if (len > 3 && w->paths[i].compare(len - 3, 3, ".gz") == 0) w->gz[i] = gzopen(paths[i], "wb1");
            else w->plain[i] = fopen(paths[i], "wb");
            if (!w->gz[i] && !w->plain[i]) fail(LCTY_ERR_INVALID_INPUT, "cannot create %s", paths[i]);
        }
        *out = w.release();
    });
}

int32_t lcty_fastx_write_recruited(lcty_fastx* f, lcty_fastx_writers* w, uint32_t max_out, const uint32_t* out_cnt, const uint32_t* out_loci,
                                   uint64_t* n_written) {
    return guarded([&] {
        if (!f || !w || !out_cnt || !out_loci || max_out == 0) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        const size_t per = f->paired ? 2 : 1;
        uint64_t written = 0;
        std::string text;
        for (uint64_t i = 0; i < f->n; i++) {
            if (!out_cnt[i]) continue;
            text.clear();
            for (size_t e = 0; e < per; e++) record_text(f->recs[i * per + e], text);       // [T; 2]::write_to: the mates one after the other
            for (uint32_t j = 0; j < std::min(out_cnt[i], max_out); j++) {
                const uint32_t locus = out_loci[static_cast<size_t>(i) * max_out + j];
                if (locus >= w->paths.size()) fail(LCTY_ERR_INVALID_INPUT, "record %llu is recruited to locus %u of %zu writers", static_cast<unsigned long long>(i), locus, w->paths.size());
                w->put(locus, text);
            }
            written++;
        }
        if (n_written) *n_written = written;
    });
}

int32_t lcty_fastx_writers_close(lcty_fastx_writers* w) {
    return guarded([&] {
        if (!w) return;
        bool ok = true;
        for (gzFile& g : w->gz) if (g) { ok &= gzclose(g) == Z_OK; g = nullptr; }
        for (FILE*& p : w->plain) if (p) { ok &= fclose(p) == 0; p = nullptr; }
        delete w;
        if (!ok) fail(LCTY_ERR_RUNTIME, "write error while closing the per-locus read files");
    });
}

}  // extern "C"
