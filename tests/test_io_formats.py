"""The file formats at the edges of the path (lcty_io.hip), host code: containers (gzip, LZ4 frames, brotli
streams), distr.gz -> lcty_bg, res.json.gz, aln.bam -> the flat alignment table. No device needed."""
import ctypes as C
import ctypes.util
import gzip
import json
import math
import os
import struct
import zlib

import numpy as np
import pytest

from locityper_amd import _lib, api, cdefs, io as lio, synth
from tests.helpers import make_bg


def _cdll(name):
    try:
        return C.CDLL(name)
    except OSError:
        return None


# ------------------------------------------------------------------ containers
def test_gzip_round_trip_and_members(tmp_path):
    data = os.urandom(70_000) + b"locityper" * 5000
    p = tmp_path / "x.bin.gz"
    lio.write_gz(p, data)
    assert gzip.open(p, "rb").read() == data and lio.read_file(p) == data
    # several members one after the other (a file written in pieces; the BGZF blocks of a BAM)
    p2 = tmp_path / "y.gz"
    p2.write_bytes(gzip.compress(data[:1000]) + gzip.compress(b"") + gzip.compress(data[1000:]))
    assert lio.read_file(p2) == data
    p3 = tmp_path / "plain.bin"
    p3.write_bytes(data)
    assert lio.read_file(p3) == data
    p4 = tmp_path / "bad.gz"
    p4.write_bytes(gzip.compress(data)[:-20])
    with pytest.raises(_lib.LocityperError) as e:
        lio.read_file(p4)
    assert e.value.code == cdefs.ERR_INVALID_DATA
    with pytest.raises(_lib.LocityperError) as e:
        lio.read_file(tmp_path / "missing.gz")
    assert e.value.code == cdefs.ERR_INVALID_INPUT


def test_bgzf_blocks_inflate_in_parallel(tmp_path):
    """BGZF from end to end (every member carries its size in a "BC" extra field): the blocks inflate independently on the host's
    cores; anything else — a plain member in between, a damaged block — takes the serial walk or fails loudly."""
    def bgzf(data, block=0xff00):
        out = bytearray()
        for i in list(range(0, len(data), block)) + [None]:
            chunk = b"" if i is None else data[i:i + block]
            comp = zlib.compressobj(6, zlib.DEFLATED, -15)
            body = comp.compress(chunk) + comp.flush()
            out += struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(body) + 25) + body
            out += struct.pack("<II", zlib.crc32(chunk), len(chunk))
        return bytes(out)
    rng = np.random.default_rng(8)
    data = bytes(rng.integers(0, 7, 6_000_000, dtype=np.uint8)) + os.urandom(200_000)          # ~100 blocks, the last ones incompressible
    blob = bgzf(data)
    p = tmp_path / "big.bgz"
    p.write_bytes(blob)
    assert lio.read_file(p) == data
    p2 = tmp_path / "small_blocks.bam"                                      # .bam: the same container
    p2.write_bytes(bgzf(data[:300_000], block=1000))
    assert lio.read_file(p2) == data[:300_000]
    p3 = tmp_path / "mixed.gz"                                              # a plain member first: not BGZF from end to end
    p3.write_bytes(gzip.compress(b"head ") + bgzf(data[:100_000]))
    assert lio.read_file(p3) == b"head " + data[:100_000]
    bad = bytearray(blob)
    bad[len(bad) // 2] ^= 0x5A                                              # inside the deflate stream of a block in the middle
    p4 = tmp_path / "bad.bgz"
    p4.write_bytes(bytes(bad))
    with pytest.raises(_lib.LocityperError) as e:
        lio.read_file(p4)
    assert e.value.code == cdefs.ERR_INVALID_DATA
    p5 = tmp_path / "cut.bgz"
    p5.write_bytes(blob[:len(blob) // 2])                                   # ends inside a block
    with pytest.raises(_lib.LocityperError) as e:
        lio.read_file(p5)
    assert e.value.code == cdefs.ERR_INVALID_DATA


def _lz4_frame_by_hand(blocks):
    """An LZ4 frame from (compressed?, bytes) blocks: FLG = version 01, block-independent off (linked blocks), no checksums."""
    out = struct.pack("<I", 0x184D2204) + bytes([0x40, 0x40, 0x00])       # FLG, BD (64 KB), HC (not checked)
    for compressed, payload in blocks:
        out += struct.pack("<I", len(payload) | (0 if compressed else 0x80000000)) + payload
    return out + struct.pack("<I", 0)


def test_lz4_frames_by_hand_and_from_liblz4(tmp_path):
    # a hand-made compressed block: 5 literals "abcde", match offset 5 length 4+7 = 11 (overlapping copy), then last literals "XYZ"
    blk = bytes([0x57]) + b"abcde" + struct.pack("<H", 5) + bytes([0x30]) + b"XYZ"
    want = b"abcde" + (b"abcde" * 3)[:11] + b"XYZ"
    p = tmp_path / "a.lz4"
    # second block refers back into the first one (block-dependent frames): offset 8 -> "deabcXYZ"[..]
    blk2 = bytes([0x10]) + b"!" + struct.pack("<H", 4) + bytes([0x00])              # literal "!", match offset 4 len 4, no last literals
    p.write_bytes(_lz4_frame_by_hand([(True, blk), (False, b"raw block"), (True, blk2[:-1])]) + struct.pack("<II", 0x184D2A50, 3) + b"skp"
                  + _lz4_frame_by_hand([(False, b"second frame")]))
    got = lio.read_file(p)
    first = want + b"raw block" + b"!"
    first += first[-4:]                                                              # the match of blk2: 4 bytes from 4 back
    assert got == first + b"second frame"
    lz4 = _cdll("liblz4.so.1")
    if lz4 is not None:
        rng = np.random.default_rng(3)
        data = bytes(rng.integers(0, 4, 300_000, dtype=np.uint8)) + b"\x00" * 100_000 + os.urandom(5000)
        lz4.LZ4F_compressFrameBound.restype = C.c_size_t
        lz4.LZ4F_compressFrameBound.argtypes = [C.c_size_t, C.c_void_p]
        lz4.LZ4F_compressFrame.restype = C.c_size_t
        lz4.LZ4F_compressFrame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        cap = lz4.LZ4F_compressFrameBound(len(data), None)
        buf = C.create_string_buffer(cap)
        n = lz4.LZ4F_compressFrame(buf, cap, data, len(data), None)
        p2 = tmp_path / "kmers.bin.lz4"
        p2.write_bytes(buf.raw[:n] * 2)                                              # two frames one after the other
        assert lio.read_file(p2) == data * 2
    bad = tmp_path / "bad.lz4"
    bad.write_bytes(b"\x00\x01\x02\x03\x04\x05\x06\x07")
    with pytest.raises(_lib.LocityperError) as e:
        lio.read_file(bad)
    assert e.value.code == cdefs.ERR_INVALID_DATA


def test_brotli_streams_through_the_system_decoder(tmp_path):
    enc = _cdll("libbrotlienc.so.1")
    if enc is None or _cdll("libbrotlidec.so.1") is None:
        pytest.skip("no system brotli on this machine")
    enc.BrotliEncoderCompress.restype = C.c_int
    enc.BrotliEncoderCompress.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p]

    def br(data, quality=5):
        cap = C.c_size_t(len(data) + 1024)
        buf = C.create_string_buffer(cap.value)
        assert enc.BrotliEncoderCompress(quality, 22, 0, len(data), data, C.byref(cap), buf) == 1
        return buf.raw[:cap.value]
    a, b = os.urandom(3000) + b"ACGT" * 50_000, b"second stream " * 1000
    p = tmp_path / "kmers.bin.br"
    p.write_bytes(br(a) + br(b))                                                     # multi-stream, as ext/sys/brotli.rs:18-86 reads
    assert lio.read_file(p) == a + b
    p.write_bytes(br(a)[:-7])
    with pytest.raises(_lib.LocityperError) as e:
        lio.read_file(p)
    assert e.value.code == cdefs.ERR_INVALID_DATA


def _stored_brotli_payload(raw):
    """The payload of a brotli stream that consists of UNCOMPRESSED meta-blocks only (RFC 7932 sections 9.1-9.2), decoded by hand: WBITS,
    then per meta-block ISLAST, MNIBBLES, MLEN - 1, ISUNCOMPRESSED, zero padding to the byte, MLEN bytes; the last one ISLAST + ISLASTEMPTY."""
    bit = 0

    def take(k):
        nonlocal bit
        v = 0
        for i in range(k):
            v |= ((raw[(bit + i) >> 3] >> ((bit + i) & 7)) & 1) << i
        bit += k
        return v
    if take(1):                                                          # WBITS: 0 = 16; else three more bits (0 there: a 7-bit form)
        assert take(3) != 0
    out = bytearray()
    while True:
        if take(1):                                                      # ISLAST
            assert take(1) == 1                                          # ISLASTEMPTY
            while bit & 7:
                assert take(1) == 0
            assert bit // 8 == len(raw)
            return bytes(out)
        nib = take(2)
        assert nib != 3
        mlen = take(4 * (nib + 4)) + 1
        assert take(1) == 1                                              # ISUNCOMPRESSED
        while bit & 7:
            assert take(1) == 0
        out += raw[bit // 8:bit // 8 + mlen]
        bit += 8 * mlen


def test_brotli_writer_stored_and_compressed(tmp_path):
    """lcty_io_write_br: the writer behind the reference's `.csv.br` tables. Stored form (quality < 0): decoded here by hand from the
    RFC's bit layout AND by the system decoder through lcty_io_read_file; compressed form: where libbrotlienc is present."""
    for data in (b"", b"x", b"stage\tgenotype\tlik\n" * 5000, os.urandom(200_001), bytes(65536), bytes(65537)):
        p = tmp_path / "sol.csv.br"
        assert lio.write_br(p, data, quality=-1) is True
        raw = p.read_bytes()
        assert _stored_brotli_payload(raw) == data
        assert len(raw) <= len(data) + 3 * (len(data) // 65536 + 1) + 1
        if _cdll("libbrotlidec.so.1") is not None:
            assert lio.read_file(p) == data
            two = tmp_path / "two.csv.br"
            two.write_bytes(raw + raw)                                   # streams one after the other, as the reference's reader takes them
            assert lio.read_file(two) == data + data
    if _cdll("libbrotlienc.so.1") is not None and _cdll("libbrotlidec.so.1") is not None:
        data = b"1\ta0,a1\t-123.4567\n" * 20000
        p = tmp_path / "reads.csv.br"
        assert lio.write_br(p, data, quality=5) is False
        assert p.stat().st_size < len(data) // 20 and lio.read_file(p) == data


def test_kmer_counts_from_compressed_files(tmp_path):
    """kmers.bin.lz4 / .gz -> lcty_io_read_file -> lcty_kmer_counts_parse (counts.rs:108-150: u8 k, u8 bytes, varint contigs, counts)."""
    def varint(v):
        out = b""
        while True:
            byte = v & 0x7F; v >>= 7
            out += bytes([byte | (0x80 if v else 0)])
            if not v:
                return out
    counts = [np.array([0, 3, 70000, 5], dtype=np.uint64), np.array([1, 0], dtype=np.uint64)]
    body = bytes([25, 2]) + varint(len(counts)) + b"".join(varint(len(c)) + b"".join(varint(int(x)) for x in c) for c in counts)
    p = tmp_path / "kmers.bin.lz4"
    p.write_bytes(_lz4_frame_by_hand([(False, body + body)]))                        # the target block follows the off-target one
    k, off, cnt, used = api.parse_kmer_counts(lio.read_file(p))
    assert (k, off.tolist(), cnt.tolist(), used) == (25, [0, 4, 6], [0, 3, 65535, 5, 1, 0], len(body))


# ------------------------------------------------------------------ distr.gz
def _distr_json(bg, read_len=150.0, tech="illumina", paired=True):
    obj = {"seq_info": {"read_len": read_len, "technology": tech, "total_reads": 123456, "file_size": None},
           "insert_distr": {"n": bg.ins_n, "p": bg.ins_p} if paired else {},
           "error_profile": {"matches": bg.op_lnprobs[0], "mismatches": bg.op_lnprobs[1], "insertions": bg.op_lnprobs[2],
                             "deletions": bg.op_lnprobs[3], "clipping": bg.op_lnprobs[4], "alpha": bg.edit_alpha, "beta": bg.edit_beta},
           "bg_depth": {"ploidy": 2, "window": bg.window, "neighb": bg.neighb, "n": list(bg.depth_n), "p": list(bg.depth_p)}}
    return obj


def test_bg_distr_from_json(tmp_path):
    bg = make_bg()
    obj = _distr_json(bg)
    p = tmp_path / "distr.gz"
    with gzip.open(p, "wt") as f:
        json.dump(obj, f, indent=4)                                                  # the reference writes it pretty-printed
    got, rl = lio.bg_from_json(lio.read_file(p))
    assert rl == 150.0 and got.technology == cdefs.TECH_ILLUMINA and got.is_paired == 1
    for f, _ in cdefs.Bg._fields_:
        a, b = getattr(got, f), getattr(bg, f)
        if hasattr(a, "__len__"):
            assert list(a) == list(b), f
        elif f != "_pad0":
            assert a == b, f
    # single-end long reads: {} insert distribution, p-value edit thresholds (EditThresh::default_for)
    bg2 = make_bg(technology=cdefs.TECH_NANOPORE, paired=False, window=3000, neighb=3000)
    got2, _ = lio.bg_from_json(json.dumps(_distr_json(bg2, 9000.5, "ONT", paired=False)))
    assert (got2.technology, got2.is_paired, got2.edit_kind, got2.edit_p1, got2.edit_p2, got2.window) == (cdefs.TECH_NANOPORE, 0, cdefs.EDIT_PVALUE, 0.99, 0.999, 3000)
    # errors of JsonSer::load
    for breaker in (lambda o: o.pop("error_profile"), lambda o: o["seq_info"].update(technology="sanger"), lambda o: o["bg_depth"]["n"].pop(),
                    lambda o: o["error_profile"].update(alpha="x"), lambda o: o.pop("bg_depth")):
        o = _distr_json(bg)
        breaker(o)
        with pytest.raises(_lib.LocityperError) as e:
            lio.bg_from_json(json.dumps(o))
        assert e.value.code == cdefs.ERR_INVALID_DATA
    with pytest.raises(_lib.LocityperError):
        lio.bg_from_json('{"seq_info": ')


# ------------------------------------------------------------------ res.json.gz
def test_res_json_is_what_into_csv_reads(tmp_path):
    call = cdefs.Call()
    call.n_out = 3
    for i, (ix, lp) in enumerate([(7, math.log(0.9)), (2, math.log(0.09)), (11, math.log(0.01))]):
        call.ixs[i] = ix; call.ln_probs[i] = lp
    call.quality = 10.0; call.unexpl_reads = 17; call.n_good = 9000; call.warnings = cdefs.WARN_FEW_READS
    gts = np.array([[1, 3], [0, 2], [2, 2]], dtype=np.uint16)
    names = ["HG001.1", "HG002.2", "NA12878", "chm13"]
    text = lio.res_to_json(call, gts, names, [-1000.0, -1010.0, -1020.0], [4.0, float("nan"), 9.0], distances=[0, 12, 0xFFFFFFFF],
                           weighted_dist=1.25)
    res = json.loads(text)
    assert list(res) == ["total_reads", "quality", "dist_type", "weight_dist", "unexpl_reads", "genotype", "options", "warnings"]
    assert res["genotype"] == "HG002.2,chm13" and res["total_reads"] == 9000 and res["unexpl_reads"] == 17 and res["dist_type"] == "minim-div"
    o0, o1, o2 = res["options"]
    assert list(o0) == ["genotype", "lik_mean", "lik_sd", "prob", "log10_prob", "dist_to_primary"]
    assert o0["lik_mean"] == -1000.0 / math.log(10.0) and o0["lik_sd"] == 4.0 / math.log(10.0)      # quirk 4: log10-scaled VARIANCE
    assert o1["lik_sd"] is None and o1["dist_to_primary"] == 12 and o2["dist_to_primary"] == "unknown"
    assert abs(o0["prob"] - 0.9) < 1e-15 and o2["genotype"] == "NA12878,NA12878"
    assert res["warnings"] == ["FewReads(9000)"]
    assert text.startswith('{\n    "total_reads": 9000,\n    "quality": 10,\n') and text.endswith("\n}")
    # extra/into_csv.py:88-96 reads exactly these
    qual = math.floor(10 * float(res["quality"])) * 0.1
    line = f"{res['genotype']}\t{qual:.1f}\t{res['total_reads']}\t{res['unexpl_reads']}\t{res['weight_dist']:.5f}\t{';'.join(res.get('warnings', '*'))}"
    assert line == "HG002.2,chm13\t10.0\t9000\t17\t1.25000\tFewReads(9000)"
    p = tmp_path / "res.json.gz"
    lio.write_gz(p, text.encode())
    assert json.load(gzip.open(p, "rt")) == res
    # no distances, no warnings
    call.warnings = 0
    res2 = json.loads(lio.res_to_json(call, gts, names, [-1.0, -2.0, -3.0], [1.0, 1.0, 1.0]))
    assert "dist_type" not in res2 and "weight_dist" not in res2 and "warnings" not in res2 and "dist_to_primary" not in res2["options"][0]


# ------------------------------------------------------------------ aln.bam
NT16 = "=ACMGRSVTWYHKDBN"


def _bgzf(data):
    """BGZF blocks (gzip members with the BC extra field) + the empty EOF block."""
    out = b""
    for i in list(range(0, len(data), 60000)) + [None]:
        chunk = b"" if i is None else data[i:i + 60000]
        comp = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = comp.compress(chunk) + comp.flush()
        bsize = len(body) + 25
        out += struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize) + body + struct.pack("<II", zlib.crc32(chunk), len(chunk))
    return out


def _bam_bytes(names_lens, records):
    """records: (qname, flag, ref_id, pos, cigar words, seq str, qual bytes|None)."""
    text = b"@HD\tVN:1.6\tSO:unsorted\n"
    out = b"BAM\x01" + struct.pack("<I", len(text)) + text + struct.pack("<I", len(names_lens))
    for nm, ln in names_lens:
        out += struct.pack("<I", len(nm) + 1) + nm.encode() + b"\0" + struct.pack("<I", ln)
    for qname, flag, ref_id, pos, cigar, seq, qual in records:
        packed = bytearray((len(seq) + 1) // 2)
        for k, ch in enumerate(seq):
            packed[k >> 1] |= NT16.index(ch) << (4 if k % 2 == 0 else 0)
        q = bytes([255] * len(seq)) if qual is None else qual
        body = struct.pack("<iiBBHHHIiii", ref_id, pos, len(qname) + 1, 37, 4680, len(cigar), flag, len(seq), -1, -1, 0)
        body += qname.encode() + b"\0" + b"".join(struct.pack("<I", w) for w in cigar) + bytes(packed) + q + b"NMC\x03"
        out += struct.pack("<I", len(body)) + body
    return out


def _unpack(chunk, mate):
    off, ln = int(chunk.mate_off[mate]), int(chunk.mate_len[mate])
    idx = off + np.arange(ln)
    codes = (chunk.bases2[idx >> 4] >> (2 * (idx & 15))) & 3
    isn = (chunk.nmask[idx >> 5] >> (idx & 31)) & 1
    return "".join("N" if n else "ACGT"[c] for c, n in zip(codes, isn))


def test_bam_reader_reproduces_the_flat_table(tmp_path):
    """A synthetic chunk written as a BAM file (BGZF, the record order of the mapper's output) and read back must be the chunk."""
    L = synth.SynthLocus(6, 300, base_len=8000)
    ch = L.reads(0, 300)
    names = [f"a{a}" for a in range(6)]
    lens = [int(L.seq_off[a + 1] - L.seq_off[a]) for a in range(6)]
    recs = []
    for r in range(ch.n_pairs):
        lo, hi = int(ch.aln_off[r]), int(ch.aln_off[r + 1])
        cbase = int(ch.cigar_off[r])
        end = 0
        for t in range(lo, hi):
            rec = ch.recs[t]
            primary = (int(rec["flags"]) & (cdefs.FLAG_SECONDARY | cdefs.FLAG_SUPPL)) == 0
            if primary and t > lo:
                end = 1
            words = ch.cigar[cbase + int(rec["cigar_rel"]): cbase + int(rec["cigar_rel"]) + int(rec["n_cigar"])].tolist()
            seq = _unpack(ch, 2 * r + end) if primary else "*"[:0]
            recs.append((f"read{r}", int(rec["flags"]), int(rec["contig"]), int(rec["pos"]), words, seq, None))
    p = tmp_path / "aln.bam"
    p.write_bytes(_bgzf(_bam_bytes(list(zip(names, lens)), recs)))
    T = lio.BamTable(p, names, paired=True)
    got = T.chunk
    assert T.n_pairs == ch.n_pairs and T.n_refs == 6 and T.names[:2] == ["read0", "read1"]
    for f in ("mate_len", "mate_off", "aln_off", "cigar_off", "cigar"):
        assert np.array_equal(getattr(got, f), getattr(ch, f)), f
    assert got.recs.tobytes() == ch.recs.tobytes()
    nb16, nb32 = (ch.n_bases + 15) // 16, (ch.n_bases + 31) // 32
    assert np.array_equal(got.bases2[:nb16], ch.bases2[:nb16]) and np.array_equal(got.nmask[:nb32], ch.nmask[:nb32])
    # the header may list a subset of the alleles in another order (strict_subset), names are mapped by name
    sub = [("a4", lens[4]), ("a1", lens[1])]
    recs2 = [("q", 0, 0, 10, [(50 << 4) | 7], "ACGTN" * 10, bytes(range(50))), ("q", 256, 1, 20, [(50 << 4) | 7], "", None),
             ("q", 16, 1, 30, [(48 << 4) | 7, (2 << 4) | 4], "T" * 50, None)]
    p.write_bytes(_bgzf(_bam_bytes(sub, recs2)))
    T2 = lio.BamTable(p, names, paired=True)
    assert T2.n_pairs == 1 and T2.n_refs == 2 and T2.chunk.recs["contig"].tolist() == [4, 1, 1] and T2.chunk.aln_off.tolist() == [0, 3]
    assert _unpack(T2.chunk, 0) == "ACGTN" * 10 and _unpack(T2.chunk, 1) == "T" * 50 and T2.chunk.mate_off.tolist() == [0, 64, 128]
    # single-end: every primary-led group is a read of its own
    T3 = lio.BamTable(p, names, paired=False)
    assert T3.n_pairs == 2 and T3.chunk.mate_len.tolist() == [50, 0, 50, 0] and T3.chunk.aln_off.tolist() == [0, 2, 3]
    # errors the reference raises: unknown contig, a second end with another name, a file that starts with a secondary record
    for names_bad, recs_bad, paired in ((names[:3], recs2, True),
                                        (names, [recs2[0], ("other", 0, 0, 5, [(50 << 4) | 7], "A" * 50, None)], True),
                                        (names, [recs2[1]], False), (names, [recs2[0]], True)):
        p.write_bytes(_bgzf(_bam_bytes(sub, recs_bad)))
        with pytest.raises(_lib.LocityperError) as e:
            lio.BamTable(p, names_bad, paired=paired)
        assert e.value.code == cdefs.ERR_INVALID_DATA


def test_bam_reader_refuses_damaged_headers(tmp_path):
    """Header words of the file are 32-bit: l_text / l_name / n_ref close to 2^32 must be refused (LCTY_ERR_INVALID_DATA), never
    wrap past the bounds check."""
    import struct
    names = ["a0", "a1"]
    good = _bam_bytes([("a0", 1000), ("a1", 1000)], [("q", 0, 0, 10, [(20 << 4) | 7], "ACGT" * 5, None)])
    p = tmp_path / "bad.bam"
    l_text = struct.unpack_from("<I", good, 4)[0]
    at_nref = 8 + l_text
    variants = []
    for v in (0xFFFFFFFF, 0xFFFFFFFC, 0x7FFFFFFF, len(good)):
        variants.append(good[:4] + struct.pack("<I", v) + good[8:])                                    # l_text
        variants.append(good[:at_nref + 4] + struct.pack("<I", v) + good[at_nref + 8:])                # l_name of the first reference
        variants.append(good[:at_nref] + struct.pack("<I", v) + good[at_nref + 4:])                    # n_ref
    variants.append(good[:at_nref + 2])                                                                # cut inside n_ref
    variants.append(good[:10])
    for raw in variants:
        p.write_bytes(_bgzf(raw))
        with pytest.raises(_lib.LocityperError) as e:
            lio.BamTable(p, names, paired=False)
        assert e.value.code == cdefs.ERR_INVALID_DATA
    p.write_bytes(_bgzf(good))
    assert lio.BamTable(p, names, paired=False).n_pairs == 1


def test_paf_reader_keeps_what_process_paf_keeps(tmp_path):
    """haplotypes.paf as command/genotype.rs:1131-1160 + seq/paf.rs read it: which lines become entries (hand-derived), the raw CIGAR
    words, the containers, the errors; a PAF written from the synthetic haplotype alignments comes back as they were."""
    names = ["h1", "h2", "h3"]
    lines = [
        "# a comment",
        "",
        "h1\t100\t0\t100\t+\th2\t102\t0\t102\t95\t104\t60\tNM:i:9\tcg:Z:50=1X20=2D29=",            # kept
        "h1\t100\t0\t100\t+\thX\t102\t0\t102\t95\t104\t60\tcg:Z:100=",                              # unknown target
        "hY\tnot-a-number\t0\t100\t+\th2\t102\t0\t102\t95\t104\t60\tcg:Z:100=",                     # unknown query: skipped before its numbers are read
        "h2\t102\t0\t102\t+\th2\t102\t0\t102\t102\t102\t60\tcg:Z:102=",                             # self-alignment
        "h1\t100\t0\t100\t+\th3\t90\t0\t90\t90\t100\t60\ttp:A:P",                                    # no CIGAR tag
        "h1\t100\t5\t100\t+\th3\t90\t0\t90\t90\t95\t60\tcg:Z:90=5I",                                 # does not start at 0
        "h1\t100\t0\t100\t-\th3\t90\t0\t90\t90\t100\t60\tcg:Z:90=10I",                               # reverse strand
        "h3\t90\t0\t90\t+\th1\t100\t0\t100\t88\t101\t60\tcg:Z:40=1X49=10D\r",                       # kept (query id > target id; CR LF)
        "h2\t102\t0\t102\t+\th3\t90\t0\t90\t0\t0\t60\tcg:Z:90M12I",                                  # kept: M passes the reader (aln_len 0: the divergence filter is lcty_locus_set_hap_alns')
    ]
    text = ("\n".join(lines) + "\n").encode()
    want = [(0, 1, [(50, 7), (1, 8), (20, 7), (2, 2), (29, 7)], 95, 104),
            (2, 0, [(40, 7), (1, 8), (49, 7), (10, 2)], 88, 101),
            (1, 2, [(90, 0), (12, 1)], 0, 0)]
    import gzip as _gz
    for fname, blob in (("haplotypes.paf", text), ("haplotypes.paf.gz", _gz.compress(text))):
        p = tmp_path / fname
        p.write_bytes(blob)
        got = lio.paf_read(p, names)
        assert [(a, b, [(int(w) >> 4, int(w) & 15) for w in ws], nm, al) for a, b, ws, nm, al in got] == want
    # contig_distances (genotype.rs:1139-1150): aln_len - n_matches of every entry with a CIGAR between two contigs of the locus,
    # whatever it covers, a later line replacing an earlier one: h1-h2 104 - 95; h1-h3 95 - 90, then 100 - 90, then (h3-h1) 101 - 88;
    # h2-h3 has aln_len 0: no distance
    _, dist = lio.paf_read(p, names, with_distances=True)
    NONE = 0xFFFFFFFF
    assert dist.tolist() == [[NONE, 9, 13], [9, NONE, NONE], [13, NONE, NONE]]
    for bad, code in (("h1\t100\t0\t100\t+\th2\t102\t0\t102\t95", cdefs.ERR_INVALID_INPUT),                              # too few columns
                      ("h1\t100\t0\tx\t+\th2\t102\t0\t102\t95\t104\t60\tcg:Z:100=", cdefs.ERR_INVALID_DATA),          # a number that does not parse
                      ("h1\t100\t0\t100\t*\th2\t102\t0\t102\t95\t104\t60\tcg:Z:100=", cdefs.ERR_INVALID_DATA),       # strand
                      ("h1\t100\t0\t100\t+\th2\t102\t0\t102\t95\t104\t60\tcg:Z:50=3N50=", cdefs.ERR_RUNTIME),        # N is refused by name
                      ("h1\t100\t0\t100\t+\th2\t102\t0\t102\t95\t104\t60\tcg:Z:50=3Q50=", cdefs.ERR_INVALID_DATA)):
        p = tmp_path / "bad.paf"
        p.write_text(bad + "\n")
        with pytest.raises(_lib.LocityperError) as e:
            lio.paf_read(p, names)
        assert e.value.code == code, bad
    # the synthetic locus: its haplotype alignments written as a PAF and read back
    from locityper_amd import synth
    L = synth.SynthLocus(5, 10, seed=3, base_len=3000)
    ents = L.hap_alns()
    hn = [f"hap{i}" for i in range(5)]
    lens = np.diff(L.seq_off).astype(int)
    out = []
    for a, b, ws, nm, al in ents:
        cg = "".join(f"{int(w) >> 4}{'MIDNSHP=X'[int(w) & 15]}" for w in ws)
        out.append(f"{hn[a]}\t{lens[a]}\t0\t{lens[a]}\t+\t{hn[b]}\t{lens[b]}\t0\t{lens[b]}\t{nm}\t{al}\t60\tcg:Z:{cg}")
    p = tmp_path / "synth.paf.gz"
    p.write_bytes(_gz.compress(("\n".join(out) + "\n").encode()))
    back = lio.paf_read(p, hn)
    assert len(back) == len(ents) > 0
    for (a, b, ws, nm, al), (a2, b2, ws2, nm2, al2) in zip(ents, back):
        assert (a, b, nm, al) == (a2, b2, nm2, al2) and np.array_equal(np.asarray(ws, dtype=np.uint32), ws2)


def test_distances_bin():
    """distances.bin (write_divergences, seq/minim_div.rs:112-124): u8 k, u8 w, varint n, the pairs i < j row by row."""
    def varint(v):
        out = bytearray()
        while True:
            b = v & 0x7F; v >>= 7
            out.append(b | (0x80 if v else 0))
            if not v: return bytes(out)
    vals = [3, 300, 70000, 0, 5, 129]                                     # (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
    blob = bytes([15, 10]) + varint(4) + b"".join(varint(v) for v in vals)
    k, w, dist = lio.distances_parse(blob, 4)
    N = 0xFFFFFFFF
    assert (k, w) == (15, 10) and dist.tolist() == [[N, 3, 300, 70000], [3, N, 0, 5], [300, 0, N, 129], [70000, 5, 129, N]]
    for bad, n in ((blob, 5), (blob[:-1], 4), (bytes([15, 10]) + varint(4) + b"\xff" * 6, 4)):
        with pytest.raises(_lib.LocityperError) as e:
            lio.distances_parse(bad, n)
        assert e.value.code == cdefs.ERR_INVALID_DATA
