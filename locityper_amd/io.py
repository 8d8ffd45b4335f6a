"""ctypes wrappers of the file-format entry points of liblocityper_hip.so (lcty_io.hip): harness code for tests, bench and examples."""
import ctypes as C

import numpy as np

from . import cdefs
from ._lib import lib, check, VP, U32, U64, D


def read_file(path):
    """lcty_io_read_file: the bytes of a file with its .gz / .lz4 / .br container removed."""
    data, n = VP(), U64()
    check(lib().lcty_io_read_file(str(path).encode(), C.byref(data), C.byref(n)))
    try:
        return C.string_at(data, n.value)
    finally:
        lib().lcty_io_free(data)


def write_gz(path, data):
    buf = bytes(data)
    check(lib().lcty_io_write_gz(str(path).encode(), buf, len(buf)))


def write_br(path, data, quality=5):
    """lcty_io_write_br: a brotli stream of `data` (the reference's `.csv.br` debug tables); returns True when the stream is STORED in
    uncompressed meta-blocks (no libbrotlienc on this system, or quality < 0)."""
    buf = bytes(data)
    stored = C.c_int32(0)
    check(lib().lcty_io_write_br(str(path).encode(), buf, len(buf), quality, C.byref(stored)))
    return bool(stored.value)


def distances_parse(data, n_alleles):
    """lcty_distances_parse: distances.bin -> (k, w, symmetric u32 matrix, NONE_U32 on the diagonal)."""
    buf = bytes(data)
    k, w = U32(), U32()
    dist = np.zeros((n_alleles, n_alleles), dtype=np.uint32)
    check(lib().lcty_distances_parse(buf, len(buf), n_alleles, C.byref(k), C.byref(w), dist.ctypes.data))
    return int(k.value), int(w.value), dist


def paf_read(path, names, with_distances=False):
    """lcty_paf_read: haplotypes.paf[.gz|.br|.lz4] -> [(id1 query, id2 target, raw CIGAR words, n_matches, aln_len)] in file order,
    the list Locus.set_hap_alns takes. names: the contig names of the locus in id order."""
    arr = (C.c_char_p * len(names))(*[n.encode() if isinstance(n, str) else bytes(n) for n in names])
    ne, nc = U64(0), U64(0)
    check(lib().lcty_paf_read(str(path).encode(), arr, len(names), C.byref(ne), None, None, None, None, None, None, C.byref(nc), None))
    n, w = int(ne.value), int(nc.value)
    id1 = np.zeros(max(n, 1), dtype=np.uint32); id2 = np.zeros(max(n, 1), dtype=np.uint32)
    nm = np.zeros(max(n, 1), dtype=np.uint32); al = np.zeros(max(n, 1), dtype=np.uint32)
    off = np.zeros(n + 1, dtype=np.uint64); words = np.zeros(max(w, 1), dtype=np.uint32)
    dist = np.zeros((len(names), len(names)), dtype=np.uint32)
    check(lib().lcty_paf_read(str(path).encode(), arr, len(names), C.byref(ne), id1.ctypes.data, id2.ctypes.data, nm.ctypes.data, al.ctypes.data,
                              off.ctypes.data, words.ctypes.data, C.byref(nc), dist.ctypes.data if with_distances else None))
    ents = [(int(id1[t]), int(id2[t]), words[int(off[t]):int(off[t + 1])].copy(), int(nm[t]), int(al[t])) for t in range(n)]
    return (ents, dist) if with_distances else ents


def bg_from_json(text):
    """BgDistr::load -> (Bg, mean read length)."""
    if isinstance(text, str):
        text = text.encode()
    bg, rl = cdefs.Bg(), D()
    check(lib().lcty_bg_from_json(text, len(text), C.byref(bg), C.byref(rl)))
    return bg, float(rl.value)


def res_to_json(call, genotypes, names, lik_mean, lik_var, distances=None, true_edit=False, weighted_dist=float("nan")):
    """Genotyping::to_json as text. genotypes[n_out][ploidy], lik_mean / lik_var[n_out] (natural log), in call.ixs order."""
    gt = np.ascontiguousarray(genotypes, dtype=np.uint16)
    n_out, ploidy = gt.shape
    assert n_out == int(call.n_out)
    nm = (C.c_char_p * len(names))(*[s.encode() for s in names])
    lm = np.ascontiguousarray(lik_mean, dtype=np.float64); lv = np.ascontiguousarray(lik_var, dtype=np.float64)
    dist = None if distances is None else np.ascontiguousarray(distances, dtype=np.uint32)
    need = U64()
    args = (C.byref(call), gt.ctypes.data, ploidy, nm, len(names), lm.ctypes.data, lv.ctypes.data,
            None if dist is None else dist.ctypes.data, int(true_edit), float(weighted_dist))
    check(lib().lcty_res_to_json(*args, None, 0, C.byref(need)))
    buf = C.create_string_buffer(int(need.value))
    check(lib().lcty_res_to_json(*args, buf, need.value, C.byref(need)))
    return buf.value.decode()


def write_bam(path, aa, chunk, names, allele_names, genotype, attempts, read_off, counts, quals=None):
    """lcty_write_bam: the alignments of the batch `aa` (scored from `chunk`) to one genotype, with the assignment counts of
    api.assignment_counts for that genotype. names: one per read pair; quals: list of bytes per mate or None. Returns the records written."""
    hs = chunk.host_struct()
    blob = "".join(names).encode()
    noff = np.zeros(len(names) + 1, dtype=np.uint64)
    np.cumsum([len(n.encode()) for n in names], out=noff[1:])
    qoff = qbuf = None
    if quals is not None:
        qoff = np.zeros(len(quals) + 1, dtype=np.uint64)
        np.cumsum([len(q) for q in quals], out=qoff[1:])
        qbuf = np.frombuffer(b"".join(quals), dtype=np.uint8).copy()
    an = (C.c_char_p * len(allele_names))(*[s.encode() for s in allele_names])
    gt = np.ascontiguousarray(genotype, dtype=np.uint16)
    ro = np.ascontiguousarray(read_off, dtype=np.uint64); cn = np.ascontiguousarray(counts, dtype=np.uint16)
    n = U64()
    check(lib().lcty_write_bam(str(path).encode(), aa._h, C.byref(hs), noff.ctypes.data, blob, None if qoff is None else qoff.ctypes.data,
                               None if qbuf is None else qbuf.ctypes.data, an, gt.ctypes.data, len(gt), attempts, ro.ctypes.data,
                               cn.ctypes.data, C.byref(n)))
    return int(n.value)


class BamTable:
    """lcty_bam_read: aln.bam as a ReadsChunk (+ read names)."""

    def __init__(self, path, names, paired=True):
        self._h = VP()
        nm = (C.c_char_p * len(names))(*[s.encode() for s in names])
        check(lib().lcty_bam_read(str(path).encode(), nm, len(names), int(paired), C.byref(self._h)))
        view = cdefs.ReadsHost()
        noff, blob, nref = VP(), VP(), U32()
        check(lib().lcty_bam_table_view(self._h, C.byref(view), C.byref(noff), C.byref(blob), C.byref(nref)))
        self.view, self.n_refs = view, int(nref.value)
        n = int(view.n_pairs)
        self.n_pairs = n

        def arr(p, count, dt):
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(dt)), (count,)).copy() if count else np.zeros(0, dtype=dt)
        mate_off = arr(view.mate_off, 2 * n + 1, C.c_uint64)
        aln_off = arr(view.aln_off, n + 1, C.c_uint64)
        cigar_off = arr(view.cigar_off, n + 1, C.c_uint64)
        n_bases = int(mate_off[-1]) if n else 0
        recs = np.frombuffer(C.string_at(view.recs, 16 * int(aln_off[-1])), dtype=cdefs.ALN_REC_DTYPE).copy() if n else np.zeros(0, dtype=cdefs.ALN_REC_DTYPE)
        self.chunk = cdefs.ReadsChunk(arr(view.mate_len, 2 * n, C.c_uint32), mate_off, arr(view.bases2, max((n_bases + 15) // 16, 2), C.c_uint32),
                                      arr(view.nmask, max((n_bases + 31) // 32, 1), C.c_uint32), aln_off, recs, cigar_off,
                                      arr(view.cigar, int(cigar_off[-1]) if n else 0, C.c_uint32))
        no = arr(noff, n + 1, C.c_uint64)
        raw = C.string_at(blob, int(no[-1])) if n else b""
        self.names = [raw[int(no[i]):int(no[i + 1])].decode() for i in range(n)]

    def close(self):
        if self._h:
            lib().lcty_bam_table_free(self._h)
            self._h = VP()

    def __del__(self):
        self.close()


class Fastx:
    """lcty_fastx_*: FASTA / FASTQ input for recruitment (src/seq/fastx.rs readers) and the per-locus writers behind it."""

    def __init__(self, path1, path2=None, interleaved=False):
        self._h = VP()
        check(lib().lcty_fastx_open(str(path1).encode(), None if path2 is None else str(path2).encode(), int(interleaved), C.byref(self._h)))
        p = C.c_int32(0)
        check(lib().lcty_fastx_is_paired(self._h, C.byref(p)))
        self.paired = bool(p.value)

    def next(self, max_records):
        """The next chunk as a cdefs.ReadsChunk of sequence fields (views into the handle: valid until the next call), or None."""
        hs = cdefs.ReadsHost()
        n = U64(0)
        check(lib().lcty_fastx_next(self._h, max_records, C.byref(hs), C.byref(n)))
        n = int(n.value)
        if n == 0:
            return None
        def arr(ptr, count, dt):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(dt)), shape=(count,))
        mate_off = arr(hs.mate_off, 2 * n + 1, C.c_uint64)
        nb = int(mate_off[-1])
        z = np.zeros(n + 1, dtype=np.uint64)
        return cdefs.ReadsChunk(arr(hs.mate_len, 2 * n, C.c_uint32), mate_off, arr(hs.bases2, nb // 16 + 1, C.c_uint32),
                                arr(hs.nmask, nb // 32 + 1, C.c_uint32), z, np.zeros(0, dtype=cdefs.ALN_REC_DTYPE), z, np.zeros(0, dtype=np.uint32))

    def write_recruited(self, writers, cnt, loci):
        cnt = np.ascontiguousarray(cnt, dtype=np.uint32)
        loci = np.ascontiguousarray(loci, dtype=np.uint32)
        n = U64(0)
        check(lib().lcty_fastx_write_recruited(self._h, writers._h, loci.shape[1], cnt.ctypes.data, loci.ctypes.data, C.byref(n)))
        return int(n.value)

    def close(self):
        if self._h:
            lib().lcty_fastx_close(self._h)
            self._h = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FastxWriters:
    def __init__(self, paths):
        self._keep = [str(p).encode() for p in paths]
        arr = (C.c_char_p * len(self._keep))(*self._keep)
        self._h = VP()
        check(lib().lcty_fastx_writers_open(arr, len(self._keep), C.byref(self._h)))

    def close(self):
        if self._h:
            h, self._h = self._h, VP()
            check(lib().lcty_fastx_writers_close(h))
