"""Synthetic workloads of the BASELINE.json shapes (binding of synth/lcty_synth.c).

Bench / test input generator: produces numpy buffers in the lcty_reads_host layout.
"""
import ctypes as C
import os

import numpy as np

from . import cdefs
from .cdefs import ALN_REC_DTYPE, Bg, ReadsChunk

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "synth", "liblcty_synth.so")
_lib = None

# seed of SURVEY.md §8(d) / BASELINE.md; locus l uses SEED + l
SEED = 0x10C17E9E20250001


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(_LIB_PATH)
        L.synth_locus_new.restype = C.c_void_p
        L.synth_locus_new.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint64]
        L.synth_locus_free.argtypes = [C.c_void_p]
        for name, rt in (("synth_locus_seqs", C.POINTER(C.c_uint8)), ("synth_locus_seq_off", C.POINTER(C.c_uint64)),
                         ("synth_locus_counts", C.POINTER(C.c_uint16)), ("synth_locus_cnt_off", C.POINTER(C.c_uint64)),
                         ("synth_locus_bg", C.POINTER(Bg))):
            getattr(L, name).restype = rt
            getattr(L, name).argtypes = [C.c_void_p]
        L.synth_locus_true_genotype.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.synth_reads_sizes.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.synth_reads_fill.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64] + [C.c_void_p] * 7
        L.synth_hap_cigar.restype = C.c_uint32
        L.synth_hap_cigar.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


class SynthLocus:
    """Alleles + off-target k-mer counts + BgDistr of one synthetic locus."""

    def __init__(self, n_alleles, n_pairs, seed=SEED, base_len=50_000, k=25,
                 technology=cdefs.TECH_ILLUMINA, read_len=150):
        L = lib()
        self._h = L.synth_locus_new(seed, n_alleles, base_len, k, technology, read_len, n_pairs)
        if not self._h:
            raise RuntimeError("synth_locus_new failed")
        self.n_alleles, self.k, self.n_pairs = n_alleles, k, n_pairs
        self.seq_off = np.ctypeslib.as_array(L.synth_locus_seq_off(self._h), (n_alleles + 1,)).copy()
        self.seqs = np.ctypeslib.as_array(L.synth_locus_seqs(self._h), (int(self.seq_off[-1]),)).copy()
        self.cnt_off = np.ctypeslib.as_array(L.synth_locus_cnt_off(self._h), (n_alleles + 1,)).copy()
        self.counts = np.ctypeslib.as_array(L.synth_locus_counts(self._h), (int(self.cnt_off[-1]),)).copy()
        self.bg = Bg()
        C.memmove(C.byref(self.bg), L.synth_locus_bg(self._h), C.sizeof(Bg))
        gt = (C.c_uint32 * 2)()
        L.synth_locus_true_genotype(self._h, gt)
        self.true_genotype = (int(gt[0]), int(gt[1]))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().synth_locus_free(self._h)
            self._h = None

    def allele(self, a):
        return bytes(self.seqs[int(self.seq_off[a]):int(self.seq_off[a + 1])])

    def hap_alns(self):
        """Pairwise allele alignments (the content of `haplotypes.paf`): [(query, target, CIGAR words, n_matches, aln_len)]
        for every query < target, as Locus.set_hap_alns takes them."""
        L = lib()
        buf = np.zeros(4 * (int(self.seq_off[-1]) // self.n_alleles // 50 + 64), dtype=np.uint32)
        nm, ln = C.c_uint32(), C.c_uint32()
        out = []
        for q in range(self.n_alleles):
            for r in range(q + 1, self.n_alleles):
                n = L.synth_hap_cigar(self._h, q, r, buf.ctypes.data, len(buf), C.byref(nm), C.byref(ln))
                if n > len(buf):
                    buf = np.zeros(2 * n, dtype=np.uint32)
                    n = L.synth_hap_cigar(self._h, q, r, buf.ctypes.data, len(buf), C.byref(nm), C.byref(ln))
                out.append((q, r, buf[:n].copy(), int(nm.value), int(ln.value)))
        return out

    def reads(self, first, n, primaries_only=False):
        """Read pairs [first, first+n) with all their candidate alignments (primaries_only: what the mapper reports when
        the other alleles are left to alignment recovery — the first record of each read end)."""
        ch = self._reads(first, n)
        return ch.primaries() if primaries_only else ch

    def _reads(self, first, n):
        L = lib()
        mate_len = np.zeros(2 * n, dtype=np.uint32)
        rec_cnt = np.zeros(n, dtype=np.uint32)
        cig_cnt = np.zeros(n, dtype=np.uint32)
        L.synth_reads_sizes(self._h, first, n, mate_len.ctypes.data, rec_cnt.ctypes.data, cig_cnt.ctypes.data)
        mate_off = np.zeros(2 * n + 1, dtype=np.uint64)
        np.cumsum((mate_len.astype(np.uint64) + 31) // 32 * 32, out=mate_off[1:])
        aln_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(rec_cnt, out=aln_off[1:], dtype=np.uint64)
        cigar_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(cig_cnt, out=cigar_off[1:], dtype=np.uint64)
        nb = int(mate_off[-1])
        bases2 = np.zeros(max(nb // 16, 1), dtype=np.uint32)
        nmask = np.zeros(max(nb // 32, 1), dtype=np.uint32)
        recs = np.zeros(int(aln_off[-1]), dtype=ALN_REC_DTYPE)
        cigar = np.zeros(max(int(cigar_off[-1]), 1), dtype=np.uint32)
        L.synth_reads_fill(self._h, first, n, mate_off.ctypes.data, bases2.ctypes.data, nmask.ctypes.data,
                           aln_off.ctypes.data, recs.ctypes.data, cigar_off.ctypes.data, cigar.ctypes.data)
        return ReadsChunk(mate_len, mate_off, bases2, nmask, aln_off, recs, cigar_off,
                          cigar[:int(cigar_off[-1])])


# the five BASELINE.json configurations
CONFIGS = {
    1: dict(n_pairs=10_000, n_alleles=8, technology=cdefs.TECH_ILLUMINA, read_len=150),
    2: dict(n_pairs=1_000_000, n_alleles=256, technology=cdefs.TECH_ILLUMINA, read_len=150),
    3: dict(n_pairs=1_000_000, n_alleles=256, technology=cdefs.TECH_NANOPORE, read_len=10_000),
    4: dict(n_pairs=1_000_000, n_alleles=256, technology=cdefs.TECH_ILLUMINA, read_len=150, n_loci=32),
    5: dict(n_pairs=5_000_000, n_alleles=4096, technology=cdefs.TECH_ILLUMINA, read_len=150),
}


def sequencer_orientation(ch):
    """The generator's chunk holds SEQ as a BAM does (reverse-complemented where the read end's primary record is on the reverse strand).
    -> a chunk with the bases as the sequencer gave them and no records: the input of candidate generation (lcty_map_reads)."""
    from .cdefs import ALN_REC_DTYPE, FLAG_MATE2, FLAG_REVERSE, FLAG_SECONDARY, FLAG_SUPPL, ReadsChunk
    b2 = ch.bases2.copy(); nm = ch.nmask.copy()
    flags = ch.recs["flags"].astype(np.int64)
    sh2 = (2 * np.arange(16)).astype(np.uint32); sh1 = np.arange(32).astype(np.uint32)
    for pair in range(ch.n_pairs):
        lo, hi = int(ch.aln_off[pair]), int(ch.aln_off[pair + 1])
        fl = flags[lo:hi]
        for e in (0, 1):
            m = 2 * pair + e
            ln, off = int(ch.mate_len[m]), int(ch.mate_off[m])
            own = fl[((fl & FLAG_MATE2) != 0) == bool(e)]
            own = own[(own & (FLAG_SECONDARY | FLAG_SUPPL)) == 0]
            if ln == 0 or len(own) == 0 or not int(own[0]) & FLAG_REVERSE:
                continue
            idx = off + np.arange(ln, dtype=np.int64)
            bases = (ch.bases2[idx >> 4] >> (2 * (idx & 15)).astype(np.uint32)) & 3
            isn = (ch.nmask[idx >> 5] >> (idx & 31).astype(np.uint32)) & 1
            rb = np.zeros((ln + 15) // 16 * 16, dtype=np.uint32); rb[:ln] = 3 - bases[::-1]
            rn = np.zeros((ln + 31) // 32 * 32, dtype=np.uint32); rn[:ln] = isn[::-1]
            b2[off >> 4:(off >> 4) + len(rb) // 16] = np.bitwise_or.reduce(rb.reshape(-1, 16) << sh2, axis=1)
            nm[off >> 5:(off >> 5) + len(rn) // 32] = np.bitwise_or.reduce(rn.reshape(-1, 32) << sh1, axis=1)
    z = np.zeros(ch.n_pairs + 1, dtype=np.uint64)
    return ReadsChunk(ch.mate_len, ch.mate_off, b2, nm, z, np.zeros(0, dtype=ALN_REC_DTYPE), z, np.zeros(0, dtype=np.uint32))
