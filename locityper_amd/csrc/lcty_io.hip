// lcty_io.hip — the file formats at the edges of the path, host code only:
//   containers   gzip / BGZF (zlib), LZ4 frames (own decoder), brotli streams (the system's libbrotlidec, looked up at run time)
//                — what ext::sys::open / create_gzip / the brotli multi-stream reader hide (src/ext/sys.rs, src/ext/sys/brotli.rs:18-86)
//   distr.gz     BgDistr::load (src/bg/mod.rs:147-177; seq_info 349-364, insert_distr insertsz.rs:183-208, error_profile
//                err_prof.rs:307-329, bg_depth depth.rs:387-412) -> lcty_bg
//   res.json.gz  Genotyping::to_json (src/solvers/solve.rs:732-773), written as the reference does (write_pretty(.., 4), gzip)
//   aln.bam      the record stream AllAlignments::load consumes (src/model/locs.rs:405-461, 502-567, 1116-1150) -> the flat table
//                of lcty_reads_append, in the input-order contract of include/locityper_hip.h
#include <dlfcn.h>
#include <zlib.h>

#include <cmath>
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <system_error>
#include <thread>
#include <unordered_map>
#include <vector>

#include "lcty_common.hpp"

using namespace lcty;

namespace {

std::vector<uint8_t> slurp(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) fail(LCTY_ERR_INVALID_INPUT, "cannot open %s", path);
    std::vector<uint8_t> buf;
    uint8_t chunk[1 << 16];
    size_t n;
    while ((n = fread(chunk, 1, sizeof(chunk), f)) > 0) buf.insert(buf.end(), chunk, chunk + n);
    const bool bad = ferror(f) != 0;
    fclose(f);
    if (bad) fail(LCTY_ERR_RUNTIME, "read error on %s", path);
    return buf;
}

bool ends_with(const std::string& s, const char* suffix) {
    const size_t n = strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

// BGZF (the blocks of a BAM / .bgz file: gzip members of <= 64 KB whose extra field "BC" holds the member's size): the members are
// independent, so they inflate on the host's cores straight into their places (htslib does the same with its thread pool). Returns
// false when the input is not BGZF from end to end (then the serial walk below takes it).
bool inflate_bgzf(const std::vector<uint8_t>& in, const char* what, std::vector<uint8_t>& out) {
    struct Block { size_t at, size, out_at; uint32_t isize; };
    std::vector<Block> blocks;
    size_t i = 0, total = 0;
    while (i < in.size()) {
        if (i + 18 > in.size() || in[i] != 0x1f || in[i + 1] != 0x8b || in[i + 2] != 8 || !(in[i + 3] & 4)) return false;
        const uint32_t xlen = in[i + 10] | (in[i + 11] << 8);
        if (i + 12 + xlen > in.size()) return false;
        uint32_t bsize = 0;
        for (size_t x = i + 12; x + 4 <= i + 12 + xlen;) {
            const uint32_t slen = in[x + 2] | (in[x + 3] << 8);
            if (in[x] == 'B' && in[x + 1] == 'C' && slen == 2 && x + 6 <= i + 12 + xlen) bsize = (in[x + 4] | (in[x + 5] << 8)) + 1u;
            x += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || i + bsize > in.size()) return false;
        const uint32_t isize = in[i + bsize - 4] | (in[i + bsize - 3] << 8) | (in[i + bsize - 2] << 16) | (static_cast<uint32_t>(in[i + bsize - 1]) << 24);
        if (isize > 65536) return false;
        blocks.push_back(Block{i, bsize, total, isize});
        total += isize; i += bsize;
    }
    out.resize(total);
    const uint32_t n_threads = static_cast<uint32_t>(std::max<size_t>(1, std::min<size_t>({blocks.size() / 64 + 1, 32, std::thread::hardware_concurrency()})));
    std::vector<int> rc_of(n_threads, Z_OK);
    auto work = [&](uint32_t tid) {
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) { rc_of[tid] = Z_MEM_ERROR; return; }
        for (size_t b = tid; b < blocks.size(); b += n_threads) {
            const Block& B = blocks[b];
            zs.next_in = const_cast<Bytef*>(in.data() + B.at); zs.avail_in = static_cast<uInt>(B.size);
            zs.next_out = out.data() + B.out_at; zs.avail_out = B.isize;
            const int rc = inflate(&zs, Z_FINISH);
            if (rc != Z_STREAM_END || zs.avail_out != 0) { rc_of[tid] = rc == Z_STREAM_END ? Z_DATA_ERROR : (rc == Z_OK ? Z_BUF_ERROR : rc); break; }
            if (inflateReset(&zs) != Z_OK) { rc_of[tid] = Z_STREAM_ERROR; break; }
        }
        inflateEnd(&zs);
    };
    // a thread that cannot be started: its share is inflated here (every thread that did start is joined)
    std::vector<std::thread> th;
    std::vector<uint32_t> here;
    for (uint32_t tid = 1; tid < n_threads; tid++) {
        try { th.emplace_back(work, tid); }
        catch (const std::system_error&) { here.push_back(tid); }
    }
    work(0);
    for (uint32_t tid : here) work(tid);
    for (auto& x : th) x.join();
    for (int rc : rc_of) if (rc != Z_OK) fail(LCTY_ERR_INVALID_DATA, "%s: corrupt BGZF block (zlib %d)", what, rc);
    return true;
}

// gzip members one after the other (a .gz written in pieces, or the BGZF blocks of a BAM file)
std::vector<uint8_t> inflate_gzip(const std::vector<uint8_t>& in, const char* what) {
    std::vector<uint8_t> out;
    if (inflate_bgzf(in, what, out)) return out;
    out.clear();
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) fail(LCTY_ERR_RUNTIME, "zlib: inflateInit2");
    zs.next_in = const_cast<Bytef*>(in.data());
    zs.avail_in = static_cast<uInt>(std::min<size_t>(in.size(), 0x7FFFFFFFu));
    size_t fed = zs.avail_in;
    std::vector<uint8_t> chunk(1 << 18);
    for (;;) {
        zs.next_out = chunk.data(); zs.avail_out = static_cast<uInt>(chunk.size());
        const int rc = inflate(&zs, Z_NO_FLUSH);
        out.insert(out.end(), chunk.data(), chunk.data() + (chunk.size() - zs.avail_out));
        if (rc == Z_STREAM_END) {
            if (zs.avail_in == 0 && fed == in.size()) break;
            if (inflateReset(&zs) != Z_OK) { inflateEnd(&zs); fail(LCTY_ERR_RUNTIME, "zlib: inflateReset"); }
        } else if (rc != Z_OK && rc != Z_BUF_ERROR) {
            inflateEnd(&zs);
            fail(LCTY_ERR_INVALID_DATA, "%s: corrupt gzip stream (zlib %d)", what, rc);
        } else if (rc == Z_BUF_ERROR && zs.avail_in == 0 && fed == in.size()) {
            inflateEnd(&zs);
            fail(LCTY_ERR_INVALID_DATA, "%s: truncated gzip stream", what);
        }
        if (zs.avail_in == 0 && fed < in.size()) {
            const size_t more = std::min<size_t>(in.size() - fed, 0x7FFFFFFFu);
            zs.next_in = const_cast<Bytef*>(in.data() + fed); zs.avail_in = static_cast<uInt>(more); fed += more;
        }
    }
    inflateEnd(&zs);
    return out;
}

// LZ4 frame format (magic 0x184D2204; skippable frames 0x184D2A50..5F), frames one after the other. Blocks of a frame may refer
// back into the blocks before them (block-dependent mode): the frame is decoded into one contiguous buffer.
uint32_t rd32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | (static_cast<uint32_t>(p[3]) << 24); }

void lz4_block(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t frame_start, const char* what) {
    size_t i = 0;
    while (i < n) {
        const uint8_t token = src[i++];
        size_t lit = token >> 4;
        if (lit == 15) { uint8_t b; do { if (i >= n) fail(LCTY_ERR_INVALID_DATA, "%s: corrupt LZ4 block", what); b = src[i++]; lit += b; } while (b == 255); }
        if (i + lit > n) fail(LCTY_ERR_INVALID_DATA, "%s: corrupt LZ4 block", what);
        out.insert(out.end(), src + i, src + i + lit);
        i += lit;
        if (i >= n) break;                                   // the last sequence of a block has literals only
        if (i + 2 > n) fail(LCTY_ERR_INVALID_DATA, "%s: corrupt LZ4 block", what);
        const size_t offset = src[i] | (src[i + 1] << 8);
        i += 2;
        size_t len = (token & 15u) + 4;
        if ((token & 15u) == 15) { uint8_t b; do { if (i >= n) fail(LCTY_ERR_INVALID_DATA, "%s: corrupt LZ4 block", what); b = src[i++]; len += b; } while (b == 255); }
        if (offset == 0 || offset > out.size() - frame_start) fail(LCTY_ERR_INVALID_DATA, "%s: LZ4 match before the start of the frame", what);
        size_t from = out.size() - offset;
        for (size_t k = 0; k < len; k++) out.push_back(out[from + k]);       // may overlap its own output
    }
}

std::vector<uint8_t> decode_lz4(const std::vector<uint8_t>& in, const char* what) {
    std::vector<uint8_t> out;
    size_t i = 0;
    while (i < in.size()) {
        if (i + 4 > in.size()) fail(LCTY_ERR_INVALID_DATA, "%s: truncated LZ4 frame", what);
        const uint32_t magic = rd32(&in[i]);
        i += 4;
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {          // skippable frame
            if (i + 4 > in.size()) fail(LCTY_ERR_INVALID_DATA, "%s: truncated LZ4 frame", what);
            i += 4 + static_cast<size_t>(rd32(&in[i]));
            continue;
        }
        if (magic != 0x184D2204u) fail(LCTY_ERR_INVALID_DATA, "%s: not an LZ4 frame (magic %08x)", what, magic);
        if (i + 3 > in.size()) fail(LCTY_ERR_INVALID_DATA, "%s: truncated LZ4 frame", what);
        const uint8_t flg = in[i];
        if ((flg >> 6) != 1) fail(LCTY_ERR_INVALID_DATA, "%s: LZ4 frame version %u", what, flg >> 6);
        const bool block_checksum = flg & 0x10, content_size = flg & 0x08, content_checksum = flg & 0x04, dict_id = flg & 0x01;
        i += 2 + (content_size ? 8 : 0) + (dict_id ? 4 : 0) + 1;          // FLG, BD, [size], [dict id], HC
        const size_t frame_start = out.size();
        for (;;) {
            if (i + 4 > in.size()) fail(LCTY_ERR_INVALID_DATA, "%s: truncated LZ4 frame", what);
            const uint32_t bs = rd32(&in[i]);
            i += 4;
            if (bs == 0) break;                              // EndMark
            const size_t n = bs & 0x7FFFFFFFu;
            if (i + n > in.size()) fail(LCTY_ERR_INVALID_DATA, "%s: truncated LZ4 block", what);
            if (bs & 0x80000000u) out.insert(out.end(), in.begin() + i, in.begin() + i + n);
            else lz4_block(&in[i], n, out, frame_start, what);
            i += n + (block_checksum ? 4 : 0);
        }
        if (content_checksum) i += 4;
    }
    return out;
}

// brotli: the decoder of the system's libbrotlidec.so.1 (public C API of google/brotli, c/include/brotli/decode.h), looked up at
// run time; several streams one after the other as the reference's reader accepts them (ext/sys/brotli.rs:18-86)
std::vector<uint8_t> decode_brotli(const std::vector<uint8_t>& in, const char* what) {
    typedef void* (*create_fn)(void*, void*, void*);
    typedef int (*stream_fn)(void*, size_t*, const uint8_t**, size_t*, uint8_t**, size_t*);
    typedef void (*destroy_fn)(void*);
    static void* lib = dlopen("libbrotlidec.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) fail(LCTY_ERR_UNSUPPORTED, "%s: brotli needs libbrotlidec.so.1, which is not on this system (%s); use the .lz4 file", what, dlerror());
    static create_fn create = reinterpret_cast<create_fn>(dlsym(lib, "BrotliDecoderCreateInstance"));
    static stream_fn stream = reinterpret_cast<stream_fn>(dlsym(lib, "BrotliDecoderDecompressStream"));
    static destroy_fn destroy = reinterpret_cast<destroy_fn>(dlsym(lib, "BrotliDecoderDestroyInstance"));
    if (!create || !stream || !destroy) fail(LCTY_ERR_UNSUPPORTED, "%s: libbrotlidec.so.1 lacks the decoder entry points", what);
    std::vector<uint8_t> out, chunk(1 << 18);
    const uint8_t* next_in = in.data();
    size_t avail_in = in.size();
    while (avail_in > 0) {
        void* st = create(nullptr, nullptr, nullptr);
        if (!st) fail(LCTY_ERR_RUNTIME, "brotli: cannot create a decoder");
        for (;;) {
            uint8_t* next_out = chunk.data();
            size_t avail_out = chunk.size();
            const int rc = stream(st, &avail_in, &next_in, &avail_out, &next_out, nullptr);
            out.insert(out.end(), chunk.data(), chunk.data() + (chunk.size() - avail_out));
            if (rc == 1) break;                              // BROTLI_DECODER_RESULT_SUCCESS: this stream is complete
            if (rc == 3) continue;                           // NEEDS_MORE_OUTPUT
            destroy(st);
            fail(LCTY_ERR_INVALID_DATA, rc == 2 ? "%s: truncated brotli stream" : "%s: corrupt brotli stream", what);
        }
        destroy(st);
    }
    return out;
}

// ---------------------------------------------------------------- a JSON value (what the `json` crate parses for JsonSer::load)
struct Json {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false; double num = 0.0; std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;
    const Json* get(const char* key) const {
        if (kind != Obj) return nullptr;
        for (const auto& kv : obj) if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

struct JsonParser {
    const char* p; const char* end;
    size_t len;
    [[noreturn]] void bad(const char* what) const { fail(LCTY_ERR_INVALID_DATA, "JSON: %s at byte %zu", what, static_cast<size_t>(p - (end - len))); }
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
    Json value(int depth = 0) {
        if (depth > 64) bad("nesting too deep");
        ws();
        if (p >= end) bad("unexpected end");
        Json v;
        if (*p == '{') {
            v.kind = Json::Obj; p++; ws();
            if (p < end && *p == '}') { p++; return v; }
            for (;;) {
                ws();
                if (p >= end || *p != '"') bad("expected a key");
                std::string k = string();
                ws();
                if (p >= end || *p != ':') bad("expected ':'");
                p++;
                v.obj.emplace_back(std::move(k), value(depth + 1));
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == '}') { p++; return v; }
                bad("expected ',' or '}'");
            }
        }
        if (*p == '[') {
            v.kind = Json::Arr; p++; ws();
            if (p < end && *p == ']') { p++; return v; }
            for (;;) {
                v.arr.push_back(value(depth + 1));
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == ']') { p++; return v; }
                bad("expected ',' or ']'");
            }
        }
        if (*p == '"') { v.kind = Json::Str; v.str = string(); return v; }
        if (end - p >= 4 && !strncmp(p, "true", 4)) { v.kind = Json::Bool; v.b = true; p += 4; return v; }
        if (end - p >= 5 && !strncmp(p, "false", 5)) { v.kind = Json::Bool; p += 5; return v; }
        if (end - p >= 4 && !strncmp(p, "null", 4)) { p += 4; return v; }
        const char* q = p;
        while (q < end && (isdigit(static_cast<unsigned char>(*q)) || *q == '-' || *q == '+' || *q == '.' || *q == 'e' || *q == 'E')) q++;
        if (q == p) bad("unexpected character");
        const std::string num(p, q);
        char* e = nullptr;
        v.kind = Json::Num; v.num = strtod(num.c_str(), &e);
        if (*e) bad("malformed number");
        p = q;
        return v;
    }
    std::string string() {
        std::string s;
        p++;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) bad("unterminated string");
                switch (*p) {
                    case 'n': s += '\n'; break; case 't': s += '\t'; break; case 'r': s += '\r'; break;
                    case 'b': s += '\b'; break; case 'f': s += '\f'; break;
                    case 'u': {
                        if (end - p < 5) bad("unterminated string");
                        const unsigned cp = static_cast<unsigned>(strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16));
                        if (cp < 0x80) s += static_cast<char>(cp);
                        else if (cp < 0x800) { s += static_cast<char>(0xC0 | (cp >> 6)); s += static_cast<char>(0x80 | (cp & 63)); }
                        else { s += static_cast<char>(0xE0 | (cp >> 12)); s += static_cast<char>(0x80 | ((cp >> 6) & 63)); s += static_cast<char>(0x80 | (cp & 63)); }
                        p += 4; break;
                    }
                    default: s += *p;
                }
                p++;
            } else s += *p++;
        }
        if (p >= end) bad("unterminated string");
        p++;
        return s;
    }
};

double need_f64(const Json& o, const char* key, const char* where) {          // json_get!(obj => key (as_f64))
    const Json* v = o.get(key);
    if (!v || v->kind != Json::Num) fail(LCTY_ERR_INVALID_DATA, "%s: Failed to parse: missing or incorrect '%s' field", where, key);
    return v->num;
}

// shortest decimal text that reads back as the same f64; NaN / inf become null as the `json` crate writes them
std::string json_num(double x) {
    if (!std::isfinite(x)) return "null";
    char buf[64];
    int prec = 1;
    for (; prec <= 17; prec++) {
        snprintf(buf, sizeof(buf), "%.*g", prec, x);
        if (strtod(buf, nullptr) == x) break;
    }
    if (strchr(buf, 'e') && x != 0.0 && std::fabs(x) >= 1e-5 && std::fabs(x) < 1e17) {      // positional notation for ordinary magnitudes
        const int e10 = static_cast<int>(std::floor(std::log10(std::fabs(x))));
        snprintf(buf, sizeof(buf), "%.*f", std::max(0, prec - 1 - e10), x);
        if (strtod(buf, nullptr) != x) snprintf(buf, sizeof(buf), "%.*g", prec, x);
    }
    return buf;
}
std::string json_str(const std::string& s) {
    std::string o = "\"";
    for (const char c : s) {
        if (c == '"' || c == '\\') { o += '\\'; o += c; }
        else if (c == '\n') o += "\\n";
        else if (c == '\t') o += "\\t";
        else if (static_cast<unsigned char>(c) < 0x20) { char b[8]; snprintf(b, sizeof(b), "\\u%04x", c); o += b; }
        else o += c;
    }
    return o + "\"";
}

}  // namespace

// aln.bam as the flat table of the boundary (+ what only file writers need: names, qualities)
struct lcty_bam_table {
    std::vector<uint32_t> mate_len; std::vector<uint64_t> mate_off; std::vector<uint32_t> bases2, nmask;
    std::vector<uint64_t> aln_off; std::vector<lcty_aln_rec> recs; std::vector<uint64_t> cigar_off; std::vector<uint32_t> cigar;
    std::vector<uint64_t> name_off; std::string names;
    std::vector<uint8_t> quals; std::vector<uint64_t> qual_off;       // per mate, BAM orientation (0xFF: absent)
    std::vector<uint8_t> mapq;                                        // per record
    uint32_t n_refs = 0;
    lcty_reads_host view{};
};

namespace {

// A brotli stream (RFC 7932) for the files the reference writes through its brotli writer — `sol.csv.br`, `reads.csv.br` and the other
// --debug tables (ext/sys.rs create_brotli; solvers/solve.rs:937-938, model/locs.rs:1062-1065). With the system's libbrotlienc.so.1
// (looked up at run time, like the decoder) the stream is compressed at `quality` (0..11; the reference's writer uses its crate's
// default); without it — or with quality < 0 — the bytes go into UNCOMPRESSED meta-blocks (RFC 7932 §9.2: ISLAST 0, MNIBBLES, MLEN - 1,
// ISUNCOMPRESSED 1, padding, the bytes; at most 65 536 bytes each so that MLEN - 1 takes four nibbles; a last empty meta-block): a
// valid stream every brotli reader takes, the reference's multi-stream reader included — stored, not compressed.
std::vector<uint8_t> brotli_stored(const uint8_t* data, uint64_t len) {
    std::vector<uint8_t> out;
    out.reserve(len + 3 * (len / 65536 + 1) + 2);
    bool first = true;
    for (uint64_t at = 0; at < len;) {
        const uint32_t n = static_cast<uint32_t>(std::min<uint64_t>(len - at, 65536));
        // the stream starts with WBITS = 22 (the four bits 1011, least significant first); a meta-block header is ISLAST (0), MNIBBLES
        // (00: four nibbles), MLEN - 1 (16 bits), ISUNCOMPRESSED (1): 20 bits — 24 with WBITS in front, or with four bits of padding
        uint32_t bits = 0, nb = 0;
        auto put = [&](uint32_t v, uint32_t k) { bits |= v << nb; nb += k; };
        if (first) put(0xBu, 4);
        put(0, 1); put(0, 2); put(n - 1, 16); put(1, 1);
        out.push_back(static_cast<uint8_t>(bits)); out.push_back(static_cast<uint8_t>(bits >> 8)); out.push_back(static_cast<uint8_t>(bits >> 16));
        out.insert(out.end(), data + at, data + at + n);
        at += n; first = false;
    }
    out.push_back(first ? 0x3Bu : 0x03u);        // ISLAST 1, ISLASTEMPTY 1 (behind WBITS when the stream is empty), padding
    return out;
}

std::vector<uint8_t> brotli_encode(const uint8_t* data, uint64_t len, int32_t quality, bool* stored) {
    *stored = true;
    if (quality >= 0) {
        using compress_fn = int (*)(int, int, int, size_t, const uint8_t*, size_t*, uint8_t*);
        using bound_fn = size_t (*)(size_t);
        static void* lib = dlopen("libbrotlienc.so.1", RTLD_NOW | RTLD_LOCAL);
        static compress_fn compress = lib ? reinterpret_cast<compress_fn>(dlsym(lib, "BrotliEncoderCompress")) : nullptr;
        static bound_fn bound = lib ? reinterpret_cast<bound_fn>(dlsym(lib, "BrotliEncoderMaxCompressedSize")) : nullptr;
        if (compress && bound) {
            size_t cap = bound(static_cast<size_t>(len));
            if (cap == 0) cap = static_cast<size_t>(len) + (len >> 2) + 1024;
            std::vector<uint8_t> out(cap);
            size_t n = cap;
            if (compress(std::min(quality, 11), 22, 0, static_cast<size_t>(len), data, &n, out.data()) == 1) {
                out.resize(n);
                *stored = false;
                return out;
            }
        }
    }
    return brotli_stored(data, len);
}

}  // namespace

extern "C" {

int32_t lcty_io_read_file(const char* path, uint8_t** data, uint64_t* len) {
    return guarded([&] {
        if (!path || !data || !len) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        std::vector<uint8_t> raw = slurp(path);
        const std::string p(path);
        std::vector<uint8_t> out;
        if (ends_with(p, ".gz") || ends_with(p, ".bgz") || ends_with(p, ".bam")) out = inflate_gzip(raw, path);
        else if (ends_with(p, ".lz4")) out = decode_lz4(raw, path);
        else if (ends_with(p, ".br")) out = decode_brotli(raw, path);
        else out.swap(raw);
        uint8_t* buf = static_cast<uint8_t*>(malloc(out.size() ? out.size() : 1));
        if (!buf) throw std::bad_alloc();
        memcpy(buf, out.data(), out.size());
        *data = buf; *len = out.size();
    });
}

void lcty_io_free(void* p) { free(p); }

int32_t lcty_io_write_gz(const char* path, const uint8_t* data, uint64_t len) {
    return guarded([&] {
        if (!path || (len && !data)) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        gzFile f = gzopen(path, "wb");
        if (!f) fail(LCTY_ERR_INVALID_INPUT, "cannot create %s", path);
        uint64_t at = 0;
        while (at < len) {
            const unsigned n = static_cast<unsigned>(std::min<uint64_t>(len - at, 1u << 30));
            if (gzwrite(f, data + at, n) != static_cast<int>(n)) { gzclose(f); fail(LCTY_ERR_RUNTIME, "write error on %s", path); }
            at += n;
        }
        if (gzclose(f) != Z_OK) fail(LCTY_ERR_RUNTIME, "write error on %s", path);
    });
}

int32_t lcty_io_write_br(const char* path, const uint8_t* data, uint64_t len, int32_t quality, int32_t* stored) {
    return guarded([&] {
        if (!path || (len && !data)) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        bool st = true;
        const std::vector<uint8_t> out = brotli_encode(data, len, quality, &st);
        FILE* f = fopen(path, "wb");
        if (!f) fail(LCTY_ERR_INVALID_INPUT, "cannot create %s", path);
        const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
        if (fclose(f) != 0 || !ok) fail(LCTY_ERR_RUNTIME, "write error on %s", path);
        if (stored) *stored = st ? 1 : 0;
    });
}

// BgDistr::load (bg/mod.rs:159-177). *read_len = seq_info.read_len (mean read length; window sizes derive from it upstream)
int32_t lcty_bg_from_json(const char* text, uint64_t len, lcty_bg* bg, double* read_len) {
    return guarded([&] {
        if (!text || !bg) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        JsonParser P{text, text + len, len};
        const Json root = P.value();
        P.ws();
        if (P.p != P.end) P.bad("trailing characters");
        const Json* seq = root.get("seq_info"); const Json* ins = root.get("insert_distr"); const Json* err = root.get("error_profile");
        if (!seq || !ins || !err)
            fail(LCTY_ERR_INVALID_DATA, "BgDistr: Failed to parse: missing 'seq_info', 'insert_distr' or 'error_profile' keys!");
        memset(bg, 0, sizeof(*bg));
        // SequencingInfo::load (bg/mod.rs:359-363); Technology::from_str (273-281)
        const double rl = need_f64(*seq, "read_len", "SequencingInfo");
        if (read_len) *read_len = rl;
        const Json* tech = seq->get("technology");
        if (!tech || tech->kind != Json::Str) fail(LCTY_ERR_INVALID_DATA, "SequencingInfo: Failed to parse: missing or incorrect 'technology' field");
        std::string t = tech->str;
        for (auto& c : t) c = static_cast<char>(tolower(static_cast<unsigned char>(c)));
        if (t == "illumina" || t == "sr") bg->technology = LCTY_TECH_ILLUMINA;
        else if (t == "hifi") bg->technology = LCTY_TECH_HIFI;
        else if (t == "pacbio" || t == "pb") bg->technology = LCTY_TECH_PACBIO;
        else if (t == "nanopore" || t == "ont") bg->technology = LCTY_TECH_NANOPORE;
        else fail(LCTY_ERR_INVALID_DATA, "Unknown technology \"%s\"", tech->str.c_str());
        // InsertDistr::load (insertsz.rs:195-208): {} = single-end
        if (ins->kind != Json::Obj) fail(LCTY_ERR_INVALID_DATA, "InsertDistr: Failed to parse: not an object");
        bg->is_paired = !ins->obj.empty();
        if (bg->is_paired) { bg->ins_n = need_f64(*ins, "n", "InsertDistr"); bg->ins_p = need_f64(*ins, "p", "InsertDistr"); }
        // ErrorProfile::load (err_prof.rs:321-329)
        static const char* const ops[5] = {"matches", "mismatches", "insertions", "deletions", "clipping"};
        for (int i = 0; i < 5; i++) bg->op_lnprobs[i] = need_f64(*err, ops[i], "ErrorProfile");
        bg->edit_alpha = need_f64(*err, "alpha", "ErrorProfile"); bg->edit_beta = need_f64(*err, "beta", "ErrorProfile");
        // EditThresh::default_for (err_prof.rs:394-399)
        if (bg->technology == LCTY_TECH_ILLUMINA) { bg->edit_kind = LCTY_EDIT_FRACTION; bg->edit_p1 = 0.03; bg->edit_p2 = 0.06; }
        else { bg->edit_kind = LCTY_EDIT_PVALUE; bg->edit_p1 = 0.99; bg->edit_p2 = 0.999; }
        // ReadDepth::load (depth.rs:400-412); `locityper genotype` needs it (BgDistr::depth().expect)
        const Json* dep = root.get("bg_depth");
        if (!dep) fail(LCTY_ERR_INVALID_DATA, "BgDistr: no 'bg_depth' key: the background read depth is required for genotyping");
        (void)need_f64(*dep, "ploidy", "ReadDepth");
        bg->window = static_cast<uint32_t>(need_f64(*dep, "window", "ReadDepth"));
        bg->neighb = static_cast<uint32_t>(need_f64(*dep, "neighb", "ReadDepth"));
        for (int which = 0; which < 2; which++) {
            const char* key = which ? "p" : "n";
            const Json* a = dep->get(key);
            if (!a || a->kind != Json::Arr) fail(LCTY_ERR_INVALID_DATA, "ReadDepth: Failed to parse: missing or incorrect array '%s'", key);
            if (a->arr.size() != LCTY_GC_BINS) fail(LCTY_ERR_INVALID_DATA, "ReadDepth: Failed to parse: incorrect number of elements in array `%s`", key);
            for (size_t i = 0; i < LCTY_GC_BINS; i++) {
                if (a->arr[i].kind != Json::Num) fail(LCTY_ERR_INVALID_DATA, "ReadDepth: element #%zu of array '%s' is not a float", i, key);
                (which ? bg->depth_p : bg->depth_n)[i] = a->arr[i].num;
            }
        }
    });
}

// Genotyping::to_json (solve.rs:732-773) as text, in the reference's layout (write_pretty(.., 4), genotype.rs:1256).
// names[n_alleles]: contig names; lik_mean / lik_var: natural-log values of the reported genotypes, in call->ixs order;
// distances: NULL or one entry per reported genotype (LCTY_NONE_U32 = "unknown"); weighted_dist: NaN = absent.
int32_t lcty_res_to_json(const lcty_call* call, const uint16_t* genotypes, uint32_t ploidy, const char* const* names, uint32_t n_alleles,
                         const double* lik_mean, const double* lik_var, const uint32_t* distances, int32_t true_edit_distances,
                         double weighted_dist, char* out, uint64_t cap, uint64_t* needed) {
    return guarded([&] {
        if (!call || !genotypes || !names || !lik_mean || !lik_var || !needed || ploidy == 0) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        const double INV_LN10 = 1.0 / std::log(10.0);
        auto gt_name = [&](uint64_t i) {
            std::string s;
            for (uint32_t t = 0; t < ploidy; t++) {
                const uint16_t a = genotypes[i * ploidy + t];
                if (a >= n_alleles) fail(LCTY_ERR_INVALID_INPUT, "genotype refers to allele %u >= %u", a, n_alleles);
                if (t) s += ',';
                s += names[a];
            }
            return s;
        };
        std::string j = "{\n";
        auto field = [&](int indent, const char* key, const std::string& val, bool last = false) {
            j += std::string(static_cast<size_t>(indent), ' ') + "\"" + key + "\": " + val + (last ? "\n" : ",\n");
        };
        std::vector<std::pair<std::string, std::string>> top;
        top.emplace_back("total_reads", std::to_string(call->n_good));
        top.emplace_back("quality", json_num(call->quality));
        if (distances) top.emplace_back("dist_type", true_edit_distances ? "\"edit\"" : "\"minim-div\"");
        if (!std::isnan(weighted_dist)) top.emplace_back("weight_dist", json_num(weighted_dist));
        top.emplace_back("unexpl_reads", std::to_string(call->unexpl_reads));
        if (call->n_out) {
            top.emplace_back("genotype", json_str(gt_name(0)));
            std::string opts = "[\n";
            for (uint64_t i = 0; i < call->n_out; i++) {
                opts += "        {\n";
                std::vector<std::pair<std::string, std::string>> o;
                o.emplace_back("genotype", json_str(gt_name(i)));
                o.emplace_back("lik_mean", json_num(lik_mean[i] * INV_LN10));
                o.emplace_back("lik_sd", json_num(lik_var[i] * INV_LN10));           // as written upstream: the log10-scaled VARIANCE (solve.rs:755)
                o.emplace_back("prob", json_num(std::exp(call->ln_probs[i])));
                o.emplace_back("log10_prob", json_num(call->ln_probs[i] * INV_LN10));
                if (distances) o.emplace_back("dist_to_primary", distances[i] == LCTY_NONE_U32 ? std::string("\"unknown\"") : std::to_string(distances[i]));
                for (size_t t = 0; t < o.size(); t++)
                    opts += "            \"" + o[t].first + "\": " + o[t].second + (t + 1 < o.size() ? ",\n" : "\n");
                opts += i + 1 < call->n_out ? "        },\n" : "        }\n";
            }
            opts += "    ]";
            top.emplace_back("options", opts);
        }
        if (call->warnings) {
            std::string w = "[\n";
            std::vector<std::string> ws;
            if (call->warnings & LCTY_WARN_NO_PROBABLE_GENOTYPE) ws.push_back("NoProbableGenotype");
            if (call->warnings & LCTY_WARN_FEW_READS) ws.push_back("FewReads(" + std::to_string(call->n_good) + ")");
            for (size_t t = 0; t < ws.size(); t++) w += "        \"" + ws[t] + "\"" + (t + 1 < ws.size() ? ",\n" : "\n");
            top.emplace_back("warnings", w + "    ]");
        }
        for (size_t t = 0; t < top.size(); t++) field(4, top[t].first.c_str(), top[t].second, t + 1 == top.size());
        j += "}";
        *needed = j.size() + 1;
        if (out && cap >= *needed) memcpy(out, j.c_str(), j.size() + 1);
        else if (out) fail(LCTY_ERR_INVALID_INPUT, "output buffer too small (%llu < %llu)", static_cast<unsigned long long>(cap), static_cast<unsigned long long>(*needed));
    });
}

// The record stream of OUT/loci/<locus>/aln.bam (name-grouped, mates mapped as independent single-end reads, genotype.rs:975-977)
// as AllAlignments::load walks it (locs.rs:1116-1150): a group = a primary record and the non-primary records behind it
// (LaggedReader, 405-461); the first group of a read is ReadEnd::First, for paired-end data the next group — which must carry the
// same name (ReadData::set_name, 147-155) — is ReadEnd::Second. Reference names are mapped to allele indices through `names`
// (construct_tid_to_contig_map, 388-401: an unknown name is an error). Sequences: the primary record's SEQ as stored.
int32_t lcty_bam_read(const char* path, const char* const* names, uint32_t n_alleles, int32_t paired, lcty_bam_table** out) {
    return guarded([&] {
        if (!path || !names || !out) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        const std::vector<uint8_t> b = inflate_gzip(slurp(path), path);
        size_t i = 0;
        // i <= b.size() always holds; n is compared with what is left, so no sum can wrap (l_text / l_name / block_size are 32-bit
        // words of the file: 0xFFFFFFFF + 4 must not pass as 3)
        auto need = [&](uint64_t n) { if (n > b.size() - i) fail(LCTY_ERR_INVALID_DATA, "%s: truncated BAM", path); };
        need(12);
        if (memcmp(&b[0], "BAM\1", 4)) fail(LCTY_ERR_INVALID_DATA, "%s: not a BAM file", path);
        i = 4;
        const uint32_t l_text = rd32(&b[i]); i += 4; need(static_cast<uint64_t>(l_text) + 4); i += l_text;
        const uint32_t n_ref = rd32(&b[i]); i += 4;
        if (n_ref > (b.size() - i) / 8) fail(LCTY_ERR_INVALID_DATA, "%s: truncated BAM (%u references in %llu bytes)", path, n_ref, static_cast<unsigned long long>(b.size() - i));
        std::map<std::string, uint32_t> by_name;
        for (uint32_t a = 0; a < n_alleles; a++) by_name[names[a]] = a;
        std::vector<uint32_t> tid2contig(n_ref);
        for (uint32_t r = 0; r < n_ref; r++) {
            need(4); const uint32_t l_name = rd32(&b[i]); i += 4; need(static_cast<uint64_t>(l_name) + 4);
            const std::string nm(reinterpret_cast<const char*>(&b[i]), l_name ? l_name - 1 : 0);
            i += static_cast<size_t>(l_name) + 4;
            const auto it = by_name.find(nm);
            if (it == by_name.end()) fail(LCTY_ERR_INVALID_DATA, "Intermediate BAM file contains unexpected contigs (e.g. %s)", nm.c_str());
            tid2contig[r] = it->second;
        }
        auto T = std::make_unique<lcty_bam_table>();
        T->n_refs = n_ref;
        T->mate_off.push_back(0); T->aln_off.push_back(0); T->cigar_off.push_back(0); T->name_off.push_back(0); T->qual_off.push_back(0);
        static const char NT16[] = "=ACMGRSVTWYHKDBN";
        uint64_t base_at = 0;
        int end_expected = 0;                                  // 0: a new read starts; 1: the second end of the current pair
        bool first_record = true;
        std::string cur_name;
        auto push_mate = [&](const uint8_t* seq, const uint8_t* qual, uint32_t l_seq) {
            T->mate_len.push_back(l_seq);
            const uint64_t words16 = (base_at + l_seq + 15) / 16, words32 = (base_at + l_seq + 31) / 32;
            T->bases2.resize(std::max<size_t>(T->bases2.size(), (words16 + 1) & ~1ull), 0);
            T->nmask.resize(std::max<size_t>(T->nmask.size(), words32), 0);
            for (uint32_t k = 0; k < l_seq; k++) {
                const char c = NT16[(seq[k >> 1] >> ((~k & 1) << 2)) & 15];
                const uint64_t at = base_at + k;
                uint32_t code = 0; bool other = false;
                switch (c) { case 'A': code = 0; break; case 'C': code = 1; break; case 'G': code = 2; break; case 'T': code = 3; break; default: other = true; }
                T->bases2[at >> 4] |= code << (2 * (at & 15));
                if (other) T->nmask[at >> 5] |= 1u << (at & 31);
            }
            T->quals.insert(T->quals.end(), qual, qual + l_seq);
            T->qual_off.push_back(T->quals.size());
            base_at = (base_at + l_seq + 31) / 32 * 32;
            T->mate_off.push_back(base_at);
        };
        auto close_pair = [&] {
            if (T->mate_len.size() % 2) { T->mate_len.push_back(0); T->mate_off.push_back(base_at); T->qual_off.push_back(T->quals.size()); }
            T->aln_off.push_back(T->recs.size());
            T->cigar_off.push_back(T->cigar.size());
        };
        while (i < b.size()) {
            need(4); const uint32_t block = rd32(&b[i]); i += 4; need(block);
            if (block < 32) fail(LCTY_ERR_INVALID_DATA, "%s: corrupt BAM record", path);
            const uint8_t* r = &b[i];
            i += block;
            const int32_t ref_id = static_cast<int32_t>(rd32(r)), pos = static_cast<int32_t>(rd32(r + 4));
            const uint32_t l_read_name = r[8], mapq = r[9], n_cigar = r[12] | (r[13] << 8), flag = r[14] | (r[15] << 8), l_seq = rd32(r + 16);
            const size_t fixed = 32 + l_read_name + 4ull * n_cigar + (l_seq + 1) / 2 + l_seq;
            if (fixed > block) fail(LCTY_ERR_INVALID_DATA, "%s: corrupt BAM record", path);
            const std::string qname(reinterpret_cast<const char*>(r + 32), l_read_name ? l_read_name - 1 : 0);
            const uint8_t* cg = r + 32 + l_read_name;
            const uint8_t* seq = cg + 4ull * n_cigar;
            const uint8_t* qual = seq + (l_seq + 1) / 2;
            const bool primary = (flag & 2304u) == 0;          // is_primary, locs.rs:405-407
            if (first_record && !primary) fail(LCTY_ERR_INVALID_DATA, "First record in the BAM file is secondary/supplementary");
            first_record = false;
            if (primary) {
                if (end_expected == 1 && paired) {
                    if (qname != cur_name) fail(LCTY_ERR_INVALID_DATA, "Read %s does not have a second read end", cur_name.c_str());
                    push_mate(seq, qual, l_seq);
                    end_expected = 0;
                } else {
                    if (!T->mate_len.empty()) close_pair();
                    cur_name = qname;
                    T->names += qname; T->name_off.push_back(T->names.size());
                    push_mate(seq, qual, l_seq);
                    end_expected = paired ? 1 : 0;
                }
            }
            lcty_aln_rec rec;
            rec.pos = pos < 0 ? 0u : static_cast<uint32_t>(pos);
            if (!(flag & 4u) && (ref_id < 0 || static_cast<uint32_t>(ref_id) >= n_ref)) fail(LCTY_ERR_INVALID_DATA, "%s: mapped record without a reference", path);
            rec.contig = static_cast<uint16_t>((flag & 4u) || ref_id < 0 ? 0 : tid2contig[ref_id]);
            rec.flags = static_cast<uint16_t>(flag & (LCTY_FLAG_UNMAPPED | LCTY_FLAG_REVERSE | LCTY_FLAG_MATE2 | LCTY_FLAG_SECONDARY | LCTY_FLAG_SUPPL));
            rec.n_cigar = n_cigar;
            rec.cigar_rel = static_cast<uint32_t>(T->cigar.size() - T->cigar_off.back());
            for (uint32_t k = 0; k < n_cigar; k++) T->cigar.push_back(rd32(cg + 4 * k));
            T->recs.push_back(rec);
            T->mapq.push_back(static_cast<uint8_t>(mapq));
        }
        if (!T->mate_len.empty()) {
            if (end_expected == 1 && paired) fail(LCTY_ERR_INVALID_DATA, "Read %s does not have a second read end", cur_name.c_str());
            close_pair();
        }
        T->bases2.resize(std::max<size_t>(T->bases2.size(), 2), 0); T->nmask.resize(std::max<size_t>(T->nmask.size(), 1), 0);
        T->view.n_pairs = T->aln_off.size() - 1;
        T->view.mate_len = T->mate_len.data(); T->view.mate_off = T->mate_off.data(); T->view.bases2 = T->bases2.data(); T->view.nmask = T->nmask.data();
        T->view.aln_off = T->aln_off.data(); T->view.recs = T->recs.data(); T->view.cigar_off = T->cigar_off.data(); T->view.cigar = T->cigar.data();
        *out = T.release();
    });
}

// the table as lcty_reads_append takes it; names (optional): name_off[n_pairs + 1] into *name_blob; *n_refs: reference sequences in
// the BAM header (fewer than the locus has alleles = Params::strict_subset, locs.rs:486)
int32_t lcty_bam_table_view(const lcty_bam_table* t, lcty_reads_host* view, const uint64_t** name_off, const char** name_blob, uint32_t* n_refs) {
    return guarded([&] {
        if (!t || !view) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        *view = t->view;
        if (name_off) *name_off = t->name_off.data();
        if (name_blob) *name_blob = t->names.data();
        if (n_refs) *n_refs = t->n_refs;
    });
}

void lcty_bam_table_free(lcty_bam_table* t) { delete t; }

// DB/loci/<locus>/haplotypes.fa.gz as ContigSet::load reads it (seq/contigs.rs:295-306 via fastx): names up to the first blank,
// sequences upper-cased and concatenated. Two calls: names / seqs NULL sizes them (*n_seqs, *names_len incl. one 0 per name, *seqs_len).
int32_t lcty_fasta_read(const char* path, uint32_t* n_seqs, char* names, uint64_t* names_len, uint8_t* seqs, uint64_t* seqs_len, uint64_t* seq_off) {
    return guarded([&] {
        if (!path || !n_seqs || !names_len || !seqs_len) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        std::vector<uint8_t> raw = slurp(path);
        const std::string p(path);
        const std::vector<uint8_t> text = (ends_with(p, ".gz") || ends_with(p, ".bgz")) ? inflate_gzip(raw, path) : raw;
        std::string nm; std::vector<uint8_t> sq; std::vector<uint64_t> off{0};
        uint32_t n = 0;
        size_t i = 0;
        while (i < text.size()) {
            size_t e = i;
            while (e < text.size() && text[e] != '\n') e++;
            size_t le = e;
            if (le > i && text[le - 1] == '\r') le--;
            if (le > i && text[i] == '>') {
                if (n) off.push_back(sq.size());
                size_t q = i + 1;
                while (q < le && text[q] != ' ' && text[q] != '\t') q++;
                nm.append(reinterpret_cast<const char*>(&text[i + 1]), q - i - 1); nm.push_back(0);
                n++;
            } else if (le > i) {
                if (!n) fail(LCTY_ERR_INVALID_DATA, "%s: sequence before the first FASTA header", path);
                for (size_t q = i; q < le; q++) sq.push_back(static_cast<uint8_t>(toupper(text[q])));
            }
            i = e + 1;
        }
        if (n) off.push_back(sq.size());
        if (names && *names_len >= nm.size()) memcpy(names, nm.data(), nm.size());
        else if (names) fail(LCTY_ERR_INVALID_INPUT, "names buffer too small");
        if (seqs && *seqs_len >= sq.size()) memcpy(seqs, sq.data(), sq.size());
        else if (seqs) fail(LCTY_ERR_INVALID_INPUT, "sequence buffer too small");
        if (seq_off) memcpy(seq_off, off.data(), sizeof(uint64_t) * off.size());
        *n_seqs = n; *names_len = nm.size(); *seqs_len = sq.size();
    });
}

// DB/loci/<locus>/haplotypes.paf[.gz|.br|.lz4] as process_paf reads it (command/genotype.rs:1131-1160; PafFile::next, seq/paf.rs:31-56;
// PafEntry::parse, 103-144): the entries lcty_locus_set_hap_alns takes. Left out, as there: empty lines and lines that start with '#',
// lines that name a contig the locus does not have, self-alignments, entries without a cg:Z: tag, and entries that do not cover both
// sequences on the forward strand (HapAlns::add drops those without remembering the pair, transfer.rs:48-52). Fewer than 12 columns,
// a number that does not parse, a strand other than + / - or a CIGAR operation outside MIDSH=X: an error, as there.
// Two calls: id1 = NULL sizes it (*n_entries, *n_cigar); with buffers, *n_entries / *n_cigar hold their capacities on entry.
int32_t lcty_paf_read(const char* path, const char* const* names, uint32_t n_alleles, uint64_t* n_entries, uint32_t* id1, uint32_t* id2,
                      uint32_t* n_matches, uint32_t* aln_len, uint64_t* cigar_off, uint32_t* cigar, uint64_t* n_cigar, uint32_t* dist) {
    return guarded([&] {
        if (!path || !names || !n_entries || !n_cigar) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        const bool fill = id1 != nullptr;
        if (fill && (!id2 || !n_matches || !aln_len || !cigar_off || !cigar)) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        std::vector<uint8_t> raw = slurp(path);
        const std::string p(path);
        const std::vector<uint8_t> text = (ends_with(p, ".gz") || ends_with(p, ".bgz")) ? inflate_gzip(raw, path)
                                          : ends_with(p, ".lz4") ? decode_lz4(raw, path) : ends_with(p, ".br") ? decode_brotli(raw, path) : raw;
        std::unordered_map<std::string, uint32_t> ids;
        for (uint32_t a = 0; a < n_alleles; a++) {
            if (!names[a]) fail(LCTY_ERR_INVALID_INPUT, "null contig name");
            ids.emplace(names[a], a);
        }
        if (dist) std::fill(dist, dist + static_cast<size_t>(n_alleles) * n_alleles, LCTY_NONE_U32);
        const uint64_t cap_e = fill ? *n_entries : 0, cap_c = fill ? *n_cigar : 0;
        uint64_t ne = 0, nc = 0;
        std::vector<uint32_t> words;
        size_t i = 0;
        while (i < text.size()) {
            size_t e = i;
            while (e < text.size() && text[e] != '\n') e++;
            size_t le = e;
            if (le > i && text[le - 1] == '\r') le--;
            const size_t line0 = i;
            i = e + 1;
            if (le == line0 || text[line0] == '#') continue;
            std::vector<std::pair<size_t, size_t>> col;                       // [begin, end)
            for (size_t q = line0, b = line0;; q++) {
                if (q == le || text[q] == '\t') { col.emplace_back(b, q); b = q + 1; if (q == le) break; }
            }
            auto str = [&](size_t c) { return std::string(reinterpret_cast<const char*>(&text[col[c].first]), col[c].second - col[c].first); };
            if (col.size() < 12) fail(LCTY_ERR_INVALID_INPUT, "%s: PAF line (%s) has too few columns", path, str(0).c_str());
            const auto q_it = ids.find(str(0));
            if (q_it == ids.end()) continue;
            const auto t_it = ids.find(str(5));
            if (t_it == ids.end()) continue;
            auto num = [&](size_t c) -> uint32_t {                              // str::parse::<u32>: digits only (a leading '+' is accepted there too)
                const std::string v = str(c);
                size_t k = (!v.empty() && v[0] == '+') ? 1 : 0;
                if (k == v.size()) fail(LCTY_ERR_INVALID_DATA, "%s: could not parse PAF line of %s", path, str(0).c_str());
                uint64_t x = 0;
                for (; k < v.size(); k++) {
                    if (v[k] < '0' || v[k] > '9') fail(LCTY_ERR_INVALID_DATA, "%s: could not parse PAF line of %s", path, str(0).c_str());
                    x = x * 10 + static_cast<uint64_t>(v[k] - '0');
                    if (x > 0xFFFFFFFFull) fail(LCTY_ERR_INVALID_DATA, "%s: could not parse PAF line of %s", path, str(0).c_str());
                }
                return static_cast<uint32_t>(x);
            };
            const uint32_t qlen = num(1), qs = num(2), qe = num(3);
            const std::string strand = str(4);
            if (strand != "+" && strand != "-") fail(LCTY_ERR_INVALID_DATA, "%s: strand '%s'", path, strand.c_str());
            const uint32_t tlen = num(6), ts = num(7), te = num(8), nm = num(9), al = num(10);
            bool have_cigar = false;
            words.clear();
            for (size_t c = 12; c < col.size(); c++) {
                if (col[c].second - col[c].first < 5 || memcmp(&text[col[c].first], "cg:Z:", 5) != 0) continue;
                have_cigar = true;
                words.clear();                                                    // a later tag replaces an earlier one
                uint32_t len = 0;
                for (size_t k = col[c].first + 5; k < col[c].second; k++) {
                    const char ch = static_cast<char>(text[k]);
                    if (ch >= '0' && ch <= '9') { len = 10 * len + static_cast<uint32_t>(ch - '0'); continue; }
                    uint32_t op;
                    switch (ch) {
                        case 'M': op = 0; break; case 'I': op = 1; break; case 'D': op = 2; break; case 'S': op = 4; break;
                        case 'H': op = 5; break; case '=': op = 7; break; case 'X': op = 8; break;
                        case 'N': case 'P': fail(LCTY_ERR_RUNTIME, "CIGAR operations N and P are not supported");
                        default: fail(LCTY_ERR_INVALID_DATA, "Unexpected CIGAR operation %c", ch);
                    }
                    words.push_back((len << 4) | op);
                    len = 0;
                }
            }
            if (q_it->second == t_it->second || !have_cigar) continue;
            // contig_distances (genotype.rs:1148-1150): the edit distance of every such entry, whatever it covers; a later line replaces
            if (dist && al != 0) {
                const uint32_t d = al - nm;
                dist[static_cast<size_t>(q_it->second) * n_alleles + t_it->second] = d;
                dist[static_cast<size_t>(t_it->second) * n_alleles + q_it->second] = d;
            }
            if (strand != "+" || qs != 0 || qe != qlen || ts != 0 || te != tlen) continue;      // full_positive_alignment, paf.rs:211-215
            if (fill) {
                if (ne >= cap_e || nc + words.size() > cap_c) fail(LCTY_ERR_INVALID_INPUT, "PAF buffers too small");
                id1[ne] = q_it->second; id2[ne] = t_it->second; n_matches[ne] = nm; aln_len[ne] = al;
                cigar_off[ne] = nc;
                memcpy(cigar + nc, words.data(), words.size() * sizeof(uint32_t));
            }
            ne++; nc += words.size();
        }
        if (fill) cigar_off[ne] = nc;
        *n_entries = ne; *n_cigar = nc;
    });
}

// DB/loci/<locus>/distances.bin (write_divergences / load_divergences_and_convert, src/seq/minim_div.rs:112-149; not compressed):
// u8 k, u8 w, varint n, then the n (n - 1) / 2 numbers of non-shared minimizers of the pairs (i, j), i < j, row by row
// (TriangleMatrix::indices, src/ext/trimat.rs:15-17) as varints. dist[n_alleles x n_alleles]: symmetric, the diagonal LCTY_NONE_U32.
int32_t lcty_distances_parse(const uint8_t* buf, uint64_t len, uint32_t n_alleles, uint32_t* k, uint32_t* w, uint32_t* dist) {
    return guarded([&] {
        if (!buf || !dist) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        uint64_t at = 0;
        auto byte = [&]() -> uint8_t {
            if (at >= len) fail(LCTY_ERR_INVALID_DATA, "distances: unexpected end of data at byte %llu", static_cast<unsigned long long>(at));
            return buf[at++];
        };
        auto varint = [&]() -> uint32_t {
            uint64_t v = 0;
            for (uint32_t i = 0; i < 5; i++) {
                const uint8_t b = byte();
                v |= static_cast<uint64_t>(b & 0x7Fu) << (7 * i);
                if (!(b & 0x80u)) return static_cast<uint32_t>(v);
            }
            fail(LCTY_ERR_INVALID_DATA, "distances: a varint of more than 5 bytes at byte %llu", static_cast<unsigned long long>(at));
            return 0;
        };
        const uint32_t kk = byte(), ww = byte();
        if (k) *k = kk;
        if (w) *w = ww;
        const uint32_t n = varint();
        if (n != n_alleles)
            fail(LCTY_ERR_INVALID_DATA, "Cannot read distances: invalid number of haplotypes (expected %u, found %u)", n_alleles, n);
        std::fill(dist, dist + static_cast<size_t>(n) * n, LCTY_NONE_U32);
        for (uint32_t i = 0; i + 1 < n; i++)
            for (uint32_t j = i + 1; j < n; j++) {
                const uint32_t d = varint();
                dist[static_cast<size_t>(i) * n + j] = d; dist[static_cast<size_t>(j) * n + i] = d;
            }
    });
}

}  // extern "C"
