"""ctypes mirrors of the plain-data structs in include/locityper_hip.h.

Shared by the product binding (api.py), the synthetic-workload generator (synth.py)
and — in tests only — the oracle binding. Field order and types must match the header.
"""
import ctypes as C

import numpy as np

GC_BINS = 101
DEPTH_CACHE = 256
MAX_ALT_CN = 15
NONE_U32 = 0xFFFFFFFF

OK, ERR_INVALID_INPUT, ERR_INVALID_DATA, ERR_RUNTIME, ERR_SOLVER, ERR_UNSUPPORTED = range(6)
TECH_ILLUMINA, TECH_HIFI, TECH_PACBIO, TECH_NANOPORE = range(4)
EDIT_FRACTION, EDIT_PVALUE = 0, 1
READ_GOOD, READ_POORLY_MAPPED, READ_OUT_OF_BOUNDS, READ_FEW_KMERS = range(4)

FLAG_UNMAPPED = 0x4
FLAG_REVERSE = 0x10
FLAG_MATE2 = 0x80
FLAG_SECONDARY = 0x100
FLAG_SUPPL = 0x800

CIGAR_M, CIGAR_I, CIGAR_D, CIGAR_S, CIGAR_H, CIGAR_EQ, CIGAR_X = 0, 1, 2, 4, 5, 7, 8


class Params(C.Structure):
    """model::Params (src/model/mod.rs:64-135)."""
    _fields_ = [
        ("boundary_size", C.c_uint32),
        ("tweak", C.c_int32),
        ("lik_skew", C.c_double),
        ("prob_diff", C.c_double),
        ("unmapped_penalty", C.c_double),
        ("poor_compl", C.c_double),
        ("poor_compl_edit", C.c_double),
        ("compl_weight_bp", C.c_double),
        ("compl_weight_pow", C.c_double),
        ("kmers_weight_bp", C.c_double),
        ("kmers_weight_pow", C.c_double),
        ("min_weight", C.c_double),
        ("filt_diff", C.c_double),
        ("prob_thresh", C.c_double),
        ("alt_cn", C.c_double * MAX_ALT_CN),
        ("n_alt_cn", C.c_uint32),
        ("kmer_soft_thresh", C.c_uint16),
        ("kmer_hard_thresh", C.c_uint16),
        ("complexity_k", C.c_uint8),
        ("dont_skip", C.c_uint8),
        ("strict_subset", C.c_uint8),
        ("_pad0", C.c_uint8),
        ("threads", C.c_uint32),
    ]


class Bg(C.Structure):
    """bg::BgDistr as loaded from distr.gz (src/bg/mod.rs:147-177)."""
    _fields_ = [
        ("op_lnprobs", C.c_double * 5),
        ("edit_alpha", C.c_double),
        ("edit_beta", C.c_double),
        ("ins_n", C.c_double),
        ("ins_p", C.c_double),
        ("depth_n", C.c_double * GC_BINS),
        ("depth_p", C.c_double * GC_BINS),
        ("edit_p1", C.c_double),
        ("edit_p2", C.c_double),
        ("window", C.c_uint32),
        ("neighb", C.c_uint32),
        ("is_paired", C.c_int32),
        ("technology", C.c_int32),
        ("edit_kind", C.c_int32),
        ("_pad0", C.c_uint32),
    ]


SOLVER_GREEDY, SOLVER_ANNEAL, SOLVER_EXACT = 0, 1, 2


class Solver(C.Structure):
    """One stage's solver (src/solvers/stoch.rs)."""
    _fields_ = [
        ("kind", C.c_int32),
        ("best_start", C.c_int32),
        ("sample_size", C.c_uint32),
        ("plato_size", C.c_uint32),
        ("anneal_steps", C.c_uint32),
        ("node_limit", C.c_uint32),
        ("init_prob", C.c_double),
    ]


class GtAlnsView(C.Structure):
    """lcty_gt_alns_view: a GenotypeAlignments after apply_tweak as plain arrays (model/assgn.rs:16-36)."""
    _fields_ = [
        ("n_reads", C.c_uint64),
        ("read_ixs", C.c_void_p),
        ("ln_prob", C.c_void_p),
        ("windows", C.c_void_p),
        ("n_windows", C.c_uint32),
        ("n_contigs", C.c_uint32),
        ("window_gc", C.c_void_p),
        ("window_weight", C.c_void_p),
        ("wshifts", C.c_void_p),
        ("depth_contrib", C.c_double),
        ("aln_contrib", C.c_double),
    ]


class DepthTables(C.Structure):
    """lcty_depth_tables: the caller's own window distributions as rows of ln-probabilities over the depth."""
    _fields_ = [("n_rows", C.c_uint32), ("width", C.c_uint32), ("values", C.c_void_p), ("id", C.c_uint64)]


class AlnRec(C.Structure):
    _fields_ = [
        ("pos", C.c_uint32),
        ("contig", C.c_uint16),
        ("flags", C.c_uint16),
        ("n_cigar", C.c_uint32),
        ("cigar_rel", C.c_uint32),
    ]


ALN_REC_DTYPE = np.dtype([("pos", "<u4"), ("contig", "<u2"), ("flags", "<u2"),
                          ("n_cigar", "<u4"), ("cigar_rel", "<u4")])
assert ALN_REC_DTYPE.itemsize == C.sizeof(AlnRec) == 16


class ReadsHost(C.Structure):
    _fields_ = [
        ("n_pairs", C.c_uint64),
        ("mate_len", C.POINTER(C.c_uint32)),
        ("mate_off", C.POINTER(C.c_uint64)),
        ("bases2", C.POINTER(C.c_uint32)),
        ("nmask", C.POINTER(C.c_uint32)),
        ("aln_off", C.POINTER(C.c_uint64)),
        ("recs", C.POINTER(AlnRec)),
        ("cigar_off", C.POINTER(C.c_uint64)),
        ("cigar", C.POINTER(C.c_uint32)),
    ]


class PairAln(C.Structure):
    _fields_ = [
        ("ln_prob", C.c_double),
        ("ix1", C.c_uint32),
        ("mid1", C.c_uint32),
        ("ix2", C.c_uint32),
        ("mid2", C.c_uint32),
        ("contig", C.c_uint16),
        ("_pad", C.c_uint16 * 3),
    ]


PAIR_ALN_DTYPE = np.dtype([("ln_prob", "<f8"), ("ix1", "<u4"), ("mid1", "<u4"), ("ix2", "<u4"),
                           ("mid2", "<u4"), ("contig", "<u2"), ("_pad", "<u2", (3,))])
assert PAIR_ALN_DTYPE.itemsize == C.sizeof(PairAln) == 32


def ptr(arr, ctype):
    """Pointer of `ctype` into a C-contiguous numpy array (kept alive by the caller)."""
    if arr is None:
        return C.cast(None, C.POINTER(ctype))
    assert arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(C.POINTER(ctype))


class ReadsChunk:
    """Numpy-backed lcty_reads_host: a chunk of read pairs with all their records."""

    def __init__(self, mate_len, mate_off, bases2, nmask, aln_off, recs, cigar_off, cigar):
        self.mate_len = np.ascontiguousarray(mate_len, dtype=np.uint32)
        self.mate_off = np.ascontiguousarray(mate_off, dtype=np.uint64)
        self.bases2 = np.ascontiguousarray(bases2, dtype=np.uint32)
        self.nmask = np.ascontiguousarray(nmask, dtype=np.uint32)
        self.aln_off = np.ascontiguousarray(aln_off, dtype=np.uint64)
        self.recs = np.ascontiguousarray(recs, dtype=ALN_REC_DTYPE)
        self.cigar_off = np.ascontiguousarray(cigar_off, dtype=np.uint64)
        self.cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
        self.n_pairs = len(self.aln_off) - 1
        assert len(self.mate_len) == 2 * self.n_pairs
        assert len(self.mate_off) == 2 * self.n_pairs + 1
        assert len(self.cigar_off) == self.n_pairs + 1
        assert np.all(self.mate_off % 32 == 0)

    @property
    def n_bases(self):
        return int(self.mate_off[-1])

    def host_struct(self):
        h = ReadsHost()
        h.n_pairs = self.n_pairs
        h.mate_len = ptr(self.mate_len, C.c_uint32)
        h.mate_off = ptr(self.mate_off, C.c_uint64)
        h.bases2 = ptr(self.bases2, C.c_uint32)
        h.nmask = ptr(self.nmask, C.c_uint32)
        h.aln_off = ptr(self.aln_off, C.c_uint64)
        h.recs = C.cast(self.recs.ctypes.data, C.POINTER(AlnRec))
        h.cigar_off = ptr(self.cigar_off, C.c_uint64)
        h.cigar = ptr(self.cigar, C.c_uint32)
        return h

    # ---- construction from plain Python, for hand-written test cases ----
    @staticmethod
    def from_pairs(pairs):
        """pairs: list of dicts {"seq1": str, "seq2": str|None, "recs": [(contig, pos, flags, cigar_str)]}.

        Records must already be in the input-order contract (mate-1 group then mate-2 group);
        FLAG_MATE2 is added by the caller. cigar_str uses =XIDSHM letters.
        """
        opmap = {"M": 0, "I": 1, "D": 2, "S": 4, "H": 5, "=": 7, "X": 8, "N": 3, "P": 6}
        mate_len, mate_off, aln_off, cigar_off = [], [0], [0], [0]
        seqs, recs, cigar = [], [], []
        for p in pairs:
            for key in ("seq1", "seq2"):
                s = p.get(key) or ""
                mate_len.append(len(s))
                seqs.append(s)
                mate_off.append(mate_off[-1] + (len(s) + 31) // 32 * 32)
            rel = 0
            for contig, pos, flags, cig in p["recs"]:
                words = []
                num = ""
                for ch in cig:
                    if ch.isdigit():
                        num += ch
                    else:
                        words.append((int(num) << 4) | opmap[ch])
                        num = ""
                recs.append((pos, contig, flags, len(words), rel))
                cigar.extend(words)
                rel += len(words)
            aln_off.append(len(recs))
            cigar_off.append(len(cigar))
        nb = mate_off[-1]
        bases2 = np.zeros(max(nb // 16, 1), dtype=np.uint32)
        nmask = np.zeros(max(nb // 32, 1), dtype=np.uint32)
        code = {"A": 0, "C": 1, "G": 2, "T": 3}
        for m, s in enumerate(seqs):
            off = mate_off[m]
            for i, ch in enumerate(s):
                b = off + i
                if ch in code:
                    bases2[b >> 4] |= np.uint32(code[ch] << (2 * (b & 15)))
                else:
                    nmask[b >> 5] |= np.uint32(1 << (b & 31))
        rec_arr = np.array(recs, dtype=ALN_REC_DTYPE) if recs else np.zeros(0, dtype=ALN_REC_DTYPE)
        return ReadsChunk(mate_len, mate_off, bases2, nmask, aln_off, rec_arr, cigar_off,
                          np.array(cigar, dtype=np.uint32) if cigar else np.zeros(0, dtype=np.uint32))

    def slice(self, lo, hi):
        """Sub-chunk of pairs [lo, hi) with rebased offsets."""
        mo = self.mate_off[2 * lo:2 * hi + 1]
        ao = self.aln_off[lo:hi + 1]
        co = self.cigar_off[lo:hi + 1]
        return ReadsChunk(
            self.mate_len[2 * lo:2 * hi], mo - mo[0],
            self.bases2[int(mo[0]) // 16:max(int(mo[-1]) // 16, int(mo[0]) // 16 + 1)],
            self.nmask[int(mo[0]) // 32:max(int(mo[-1]) // 32, int(mo[0]) // 32 + 1)],
            ao - ao[0], self.recs[int(ao[0]):int(ao[-1])], co - co[0],
            self.cigar[int(co[0]):int(co[-1])])

    def counted(self, allele_len):
        """The chunk's records as lcty_aln_counted entries (uint32 x 4 per record): what count_region_operations_fast +
        limited_clipping (aln.rs:288-317) leave of every record, computed here as the caller of lcty_reads_append_counted would.
        Records with an empty CIGAR must not be in the chunk (the record path skips them; a counted batch has no way to say so)."""
        n = len(self.recs)
        try:                                                     # the C helper of the synthetic-data library does the same loop
            from . import synth
            out = np.zeros((n, 4), dtype=np.uint32)
            al = np.ascontiguousarray(allele_len, dtype=np.uint32)
            f = synth.lib().synth_count_records
            f.restype = C.c_uint64
            f.argtypes = [C.c_uint64] + [C.c_void_p] * 5 + [C.c_uint32, C.c_void_p]
            bad = f(self.n_pairs, self.aln_off.ctypes.data, self.recs.ctypes.data, self.cigar_off.ctypes.data, self.cigar.ctypes.data,
                    al.ctypes.data, len(al), out.ctypes.data)
            assert bad == 0, f"record {bad - 1} cannot be counted (unsupported operation, empty CIGAR or a count above 65535)"
            return out
        except (ImportError, OSError, AttributeError):
            pass
        pair_of = np.repeat(np.arange(self.n_pairs, dtype=np.int64), np.diff(self.aln_off.astype(np.int64)))
        start = self.cigar_off[pair_of].astype(np.int64) + self.recs["cigar_rel"].astype(np.int64)
        ncig = self.recs["n_cigar"].astype(np.int64)
        unm = (self.recs["flags"].astype(np.int64) & FLAG_UNMAPPED) != 0
        assert np.all((ncig > 0) | unm), "a mapped record with an empty CIGAR in a chunk that is to be counted"
        rec_of = np.repeat(np.arange(n, dtype=np.int64), ncig)
        first = np.cumsum(ncig) - ncig
        word_ix = start[rec_of] + (np.arange(len(rec_of), dtype=np.int64) - first[rec_of])
        w = self.cigar[word_ix].astype(np.int64)
        op, ln = w & 15, w >> 4
        is_first = (np.arange(len(rec_of)) - first[rec_of]) == 0
        is_last = (np.arange(len(rec_of)) - first[rec_of]) == ncig[rec_of] - 1
        op = np.where((op == CIGAR_H) & (is_first | is_last), CIGAR_S, op)                       # hard_to_soft (cigar.rs:309-320)
        assert np.all(np.isin(op, (CIGAR_EQ, CIGAR_X, CIGAR_I, CIGAR_D, CIGAR_S))), "unsupported CIGAR operation"
        def total(mask):
            return np.bincount(rec_of[mask], weights=ln[mask], minlength=n).astype(np.int64)
        matches, mism, ins, dele = total(op == CIGAR_EQ), total(op == CIGAR_X), total(op == CIGAR_I), total(op == CIGAR_D)
        left = total((op == CIGAR_S) & is_first); right = total((op == CIGAR_S) & is_last & ~is_first)
        pos = self.recs["pos"].astype(np.int64)
        end = pos + matches + mism + dele
        clen = np.asarray(allele_len, dtype=np.int64)[np.minimum(self.recs["contig"].astype(np.int64), len(allele_len) - 1)]
        clip = np.minimum(left, pos) + np.minimum(right, np.maximum(clen - end, 0))               # limited_clipping (aln.rs:288-296)
        assert max(matches.max(initial=0), mism.max(initial=0), ins.max(initial=0), dele.max(initial=0), clip.max(initial=0)) < 65536
        fl = self.recs["flags"].astype(np.int64)
        pos_flags = pos | np.where(fl & FLAG_REVERSE, 1 << 28, 0) | np.where(fl & (FLAG_SECONDARY | FLAG_SUPPL), 1 << 29, 0) \
            | np.where(fl & FLAG_UNMAPPED, 1 << 30, 0)
        out = np.zeros((n, 4), dtype=np.uint32)
        out[:, 0] = pos_flags
        out[:, 1] = self.recs["contig"].astype(np.int64) | (matches << 16)
        out[:, 2] = mism | (ins << 16)
        out[:, 3] = dele | (clip << 16)
        return out

    def primaries(self):
        """The same pairs with only the primary record of each read end (no secondary / supplementary records)."""
        n = len(self.aln_off) - 1
        keep = (self.recs["flags"] & (FLAG_SECONDARY | FLAG_SUPPL)) == 0
        pair_of = np.repeat(np.arange(n, dtype=np.int64), np.diff(self.aln_off.astype(np.int64)))
        recs = self.recs[keep].copy()
        pair_k = pair_of[keep]
        aln_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(np.bincount(pair_k, minlength=n), out=aln_off[1:], dtype=np.uint64)
        nc = recs["n_cigar"].astype(np.int64)
        src = self.cigar_off.astype(np.int64)[pair_k] + recs["cigar_rel"].astype(np.int64)
        ends = np.cumsum(nc)
        starts = ends - nc
        cigar_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(np.bincount(pair_k, weights=nc, minlength=n).astype(np.uint64), out=cigar_off[1:], dtype=np.uint64)
        recs["cigar_rel"] = (starts - cigar_off.astype(np.int64)[pair_k]).astype(np.uint32)
        total = int(ends[-1]) if len(ends) else 0
        gather = np.repeat(src - starts, nc) + np.arange(total, dtype=np.int64)
        cigar = self.cigar[gather] if total else np.zeros(0, dtype=np.uint32)
        return ReadsChunk(self.mate_len, self.mate_off, self.bases2, self.nmask, aln_off, recs, cigar_off, cigar)


class RecruitParams(C.Structure):      # lcty_recruit_params
    _fields_ = [("match_frac", C.c_double), ("match_length", C.c_uint32), ("thresh_kmer_count", C.c_uint16),
                ("minimizer_k", C.c_uint8), ("minimizer_w", C.c_uint8)]


class MapParams(C.Structure):          # lcty_map_params
    _fields_ = [("k", C.c_uint32), ("stride", C.c_uint32), ("min_votes", C.c_uint32), ("max_occ", C.c_uint32), ("match", C.c_int32), ("mismatch", C.c_int32),
                ("end_bonus", C.c_int32), ("min_score", C.c_int32), ("band", C.c_uint32), ("gap_open", C.c_int32), ("gap_extend", C.c_int32),
                ("route", C.c_uint32), ("chain_gap", C.c_uint32), ("chain_skew", C.c_uint32), ("chain_back", C.c_uint32)]


MAP_ROUTE_AUTO, MAP_ROUTE_SHORT, MAP_ROUTE_LONG = 0, 1, 2


WARN_NO_PROBABLE_GENOTYPE, WARN_FEW_READS = 1, 2      # lcty_call_checks


class Stage(C.Structure):                # lcty_stage
    _fields_ = [("solver", Solver), ("in_size", C.c_uint64), ("attempts", C.c_uint32), ("_pad0", C.c_uint32)]


MAX_RESULT = 50


class Call(C.Structure):                 # lcty_call
    _fields_ = [("n_out", C.c_uint64), ("ixs", C.c_uint64 * MAX_RESULT), ("ln_probs", C.c_double * MAX_RESULT),
                ("quality", C.c_double), ("unexpl_reads", C.c_uint32), ("warnings", C.c_uint32), ("n_good", C.c_uint64),
                ("kept_after_filter", C.c_uint64)]
