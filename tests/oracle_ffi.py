"""ctypes binding of oracle/_build/liblcty_oracle.so — the CPU restatement.

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py. The product package (locityper_amd/) never imports this module.
"""
import ctypes as C
import os

import numpy as np

from locityper_amd import cdefs
from locityper_amd.cdefs import Bg, Params, ReadsHost, PAIR_ALN_DTYPE, Solver

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_ROOT, "oracle", "_build", "liblcty_oracle.so")
_lib = None

D, U32, U64, VP = C.c_double, C.c_uint32, C.c_uint64, C.c_void_p


class Rng(C.Structure):
    _fields_ = [("s", C.c_uint64 * 4)]


class NBinom(C.Structure):
    _fields_ = [("n", D), ("p", D), ("lnq", D), ("lnpmf_const", D)]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} missing: run `make -C oracle`")
    L = C.CDLL(LIB_PATH)

    def sig(name, restype, *argtypes):
        f = getattr(L, name)
        f.restype = restype
        f.argtypes = list(argtypes)

    sig("orc_ln_gamma", D, D)
    sig("orc_ln_beta", D, D, D)
    sig("orc_beta_reg", D, D, D, D)
    sig("orc_ln_add", D, D, D)
    sig("orc_ln_sum", D, VP, C.c_size_t)
    sig("orc_ln_sum_init", D, VP, C.c_size_t, D)
    sig("orc_nbinom_new", NBinom, D, D)
    sig("orc_nbinom_ln_pmf", D, C.POINTER(NBinom), U32)
    sig("orc_nbinom_mode", U32, C.POINTER(NBinom))
    sig("orc_nbinom_cdf", D, C.POINTER(NBinom), U32)
    sig("orc_nbinom_quantile", D, C.POINTER(NBinom), D)
    sig("orc_insert_cache_size", C.c_size_t, C.POINTER(NBinom))
    sig("orc_betabinom_inv_cdf2", None, D, D, U32, D, D, C.POINTER(U32), C.POINTER(U32))
    sig("orc_edit_thresholds", None, C.POINTER(Bg), U32, C.POINTER(U32), C.POINTER(U32))
    sig("orc_depth_ln_pmf", D, C.POINTER(Bg), C.POINTER(Params), U32, U32)
    sig("orc_students_t_cdf", D, D, D)
    sig("orc_t_test", D, D, D, D, D, D)
    sig("orc_t_test_diffsizes", D, D, D, D, D, D, D)
    sig("orc_params_default", None, C.POINTER(Params))
    sig("orc_params_resolve", C.c_int, C.POINTER(Params), C.POINTER(Bg))
    sig("orc_kmers_u128", C.c_size_t, VP, C.c_size_t, U32, C.c_int, VP)
    sig("orc_kmers_u32", C.c_size_t, VP, C.c_size_t, U32, C.c_int, VP)
    sig("orc_complexity_counts", C.c_size_t, VP, C.c_size_t, U32, U32, VP)
    sig("orc_locus_new", VP, U32, VP, VP, VP, VP, U32, C.POINTER(Bg), C.POINTER(Params))
    sig("orc_locus_free", None, VP)
    sig("orc_locus_n_unique_kmers", U64, VP)
    sig("orc_locus_contig_info", C.c_int, VP, U32, VP, VP, VP, C.POINTER(U32), C.POINTER(U32))
    sig("orc_locus_insert_lnprob", D, VP, U32)
    sig("orc_locus_insert_penalty", D, VP)
    sig("orc_locus_window_weights", None, VP, VP)
    sig("orc_depth_table", None, VP, VP, U32, U32, VP)
    sig("orc_load", VP, VP, C.POINTER(ReadsHost), C.POINTER(C.c_int))
    sig("orc_load_mt", VP, VP, C.POINTER(ReadsHost), VP, U32, VP, C.POINTER(C.c_int))
    sig("orc_alns_free", None, VP)
    sig("orc_alns_from_arrays", VP, U64, U32, VP, VP, VP, VP, VP)
    sig("orc_alns_n_pairs", U64, VP)
    sig("orc_alns_n_good", U64, VP)
    sig("orc_alns_status", None, VP, VP, VP, VP, VP)
    sig("orc_alns_pair_alns", U64, VP, VP, VP, U64)
    sig("orc_best_aln_matrix", None, VP, VP)
    sig("orc_count_genotypes", U64, U32, U32)
    sig("orc_generate_genotypes", U64, U32, U32, VP)
    sig("orc_run_filter", None, VP, U32, U64, VP, U64, U32, VP, VP)
    sig("orc_truncate", U64, VP, VP, U64, D, U64, U64)
    # solver stages
    sig("orc_rng_seed", None, C.POINTER(Rng), U64)
    sig("orc_rng_next", U64, C.POINTER(Rng))
    sig("orc_rng_jump", None, C.POINTER(Rng))
    sig("orc_rng_long_jump", None, C.POINTER(Rng))
    sig("orc_rng_below", U64, C.POINTER(Rng), U64)
    sig("orc_rng_f64", D, C.POINTER(Rng))
    sig("orc_counter_u64", U64, U64, U64)
    sig("orc_weight_calc", D, D, D, D)
    sig("orc_window_weight", D, VP, U32, U32, C.POINTER(U32))
    sig("orc_locus_set_explicit_weights", C.c_int, VP, U32, VP, VP, VP, VP)
    sig("orc_depth_ln_prob", D, VP, U32, D, U32)
    sig("orc_gt_alns_new", VP, VP, VP, VP, U32)
    sig("orc_gt_alns_free", None, VP)
    sig("orc_gt_alns_n_reads", U64, VP)
    sig("orc_gt_alns_n_alns", U64, VP)
    sig("orc_gt_alns_n_windows", U32, VP)
    sig("orc_gt_alns_n_nontrivial", U64, VP)
    sig("orc_gt_alns_get", None, VP, VP, VP, VP, VP, VP, VP, VP)
    sig("orc_gt_alns_apply_tweak", None, VP, U64)
    sig("orc_gt_alns_window_distr", None, VP, VP, VP)
    sig("orc_gt_alns_max_aln_lik", D, VP)
    sig("orc_solver_default", None, C.POINTER(Solver), C.c_int32)
    sig("orc_solve", D, VP, C.POINTER(Solver), U64, VP, VP)
    sig("orc_locus_inject_tables", None, VP, VP, VP)
    sig("orc_locus_inject_depth_table", None, VP, U32, VP)
    sig("orc_assignment_likelihood", D, VP, VP, VP)
    sig("orc_solve_stage", None, VP, VP, VP, U64, U32, VP, C.POINTER(Solver), U32, VP, VP, VP, VP)
    sig("orc_solve_stage_mt", None, VP, VP, VP, U64, U32, VP, C.POINTER(Solver), U32, VP, VP, VP, VP, U32)
    sig("orc_call_checks", None, VP, U64, U32, VP, U32, VP, U32, VP, C.POINTER(D), C.POINTER(U32))
    sig("orc_hap_alns_new", VP, U32, U32, D)
    sig("orc_hap_alns_free", None, VP)
    sig("orc_hap_alns_add", None, VP, U32, U32, VP, U32, U32, U32)
    sig("orc_hap_alns_sort", None, VP)
    sig("orc_load_recover", VP, VP, C.POINTER(ReadsHost), VP, C.POINTER(C.c_int))
    sig("orc_transfer_one", U32, VP, U32, U32, U32, VP, U32, VP, U32, VP, U32, VP, U32, C.POINTER(U32))
    sig("orc_transfer_set_optimize", None, C.c_int)
    sig("orc_dp_align", C.c_int, VP, U32, VP, U32, C.c_int, C.c_int, VP, U32, C.POINTER(U32))
    sig("orc_assignment_counts", U64, VP, VP, VP, U32, C.POINTER(Solver), U32, VP, VP, VP)
    sig("orc_compare_two_likelihoods", D, D, D, U32, D, D, U32)
    sig("orc_discard_improbable", U64, VP, VP, VP, VP, U64, D, U64, U64)
    sig("orc_produce_result", U64, VP, VP, VP, VP, U64, D, U64, VP, VP, C.POINTER(D))
    sig("orc_count_unexplained", U32, VP, VP, U32)
    sig("orc_fast_hash64", U64, U64)
    sig("orc_canon_minimizers", C.c_size_t, VP, C.c_size_t, U32, U32, VP, VP, VP, C.c_size_t)
    sig("orc_fraction_approximate_u16", None, D, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16))
    sig("orc_targets_new", VP, C.c_uint8, C.c_uint8, D, U32, C.c_uint16)
    sig("orc_targets_free", None, VP)
    sig("orc_targets_params", None, VP, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16), C.POINTER(U32), C.POINTER(U32))
    sig("orc_targets_add_locus", U32, VP, U32, VP, VP, VP, VP, U32)
    sig("orc_targets_finalize", None, VP)
    sig("orc_targets_n_entries", C.c_size_t, VP)
    sig("orc_targets_entry", None, VP, C.c_size_t, C.POINTER(U64), C.POINTER(U32), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8))
    sig("orc_recruit", C.c_size_t, VP, VP, C.c_size_t, VP, C.c_size_t, VP, C.c_size_t)
    _lib = L
    return L


def default_params():
    p = Params()
    lib().orc_params_default(C.byref(p))
    return p


def resolve_params(p, bg):
    rc = lib().orc_params_resolve(C.byref(p), C.byref(bg))
    if rc:
        raise ValueError(f"orc_params_resolve -> {rc}")
    return p


def kmers(seq: bytes, k: int, canonical=True):
    """kmers::kmers::<u128> as Python ints."""
    n = len(seq)
    out = np.zeros((max(n + 1 - k, 0), 2), dtype=np.uint64)
    buf = np.frombuffer(seq, dtype=np.uint8)
    w = lib().orc_kmers_u128(buf.ctypes.data, n, k, int(canonical), out.ctypes.data)
    return [int(lo) | (int(hi) << 64) for lo, hi in out[:w]]


def complexity_counts(seq: bytes, k: int, w: int):
    n = len(seq)
    out = np.zeros(n - w + 1, dtype=np.uint16)
    buf = np.frombuffer(seq, dtype=np.uint8)
    m = lib().orc_complexity_counts(buf.ctypes.data, n, k, w, out.ctypes.data)
    assert m == len(out)
    return out


class OracleLocus:
    def __init__(self, seqs, seq_off, counts, cnt_off, k, bg, params):
        self.seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        self.seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
        self.counts = np.ascontiguousarray(counts, dtype=np.uint16)
        self.cnt_off = np.ascontiguousarray(cnt_off, dtype=np.uint64)
        self.n_alleles = len(self.seq_off) - 1
        self.bg, self.params, self.k = bg, params, k
        self._h = lib().orc_locus_new(self.n_alleles, self.seqs.ctypes.data, self.seq_off.ctypes.data,
                                      self.counts.ctypes.data, self.cnt_off.ctypes.data, k,
                                      C.byref(bg), C.byref(params))
        if not self._h:
            raise ValueError("orc_locus_new failed")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_locus_free(self._h)
            self._h = None

    def n_unique_kmers(self):
        return int(lib().orc_locus_n_unique_kmers(self._h))

    def set_explicit_weights(self, allele, start, end, value):
        """load_explicit_weights (windows.rs:257-317) on parsed lines; returns the status code"""
        al = np.ascontiguousarray(allele, dtype=np.uint32); st = np.ascontiguousarray(start, dtype=np.uint32)
        en = np.ascontiguousarray(end, dtype=np.uint32); va = np.ascontiguousarray(value, dtype=np.float64)
        return int(lib().orc_locus_set_explicit_weights(self._h, len(al), al.ctypes.data, st.ctypes.data, en.ctypes.data,
                                                        va.ctypes.data))

    def contig_info(self, a):
        ln = int(self.seq_off[a + 1] - self.seq_off[a])
        npos = ln - self.bg.neighb + 1
        gc = np.zeros(npos, dtype=np.uint8)
        uniq = np.zeros(npos, dtype=np.uint32)
        cc = np.zeros(npos, dtype=np.uint16)
        nw, rs = U32(), U32()
        rc = lib().orc_locus_contig_info(self._h, a, gc.ctypes.data, uniq.ctypes.data, cc.ctypes.data,
                                         C.byref(nw), C.byref(rs))
        assert rc == 0
        return gc, uniq, cc, nw.value, rs.value

    def insert_lnprob(self, sz):
        return lib().orc_locus_insert_lnprob(self._h, int(sz))

    def insert_penalty(self):
        return lib().orc_locus_insert_penalty(self._h)

    def inject_tables(self, depth_lut=None, win_weight=None):
        """Test hook: use the tables the GPU built (bit-identical solver inputs)."""
        dl = None if depth_lut is None else np.ascontiguousarray(depth_lut, dtype=np.float64)
        ww = None if win_weight is None else np.ascontiguousarray(win_weight, dtype=np.float64)
        lib().orc_locus_inject_tables(self._h, None if dl is None else dl.ctypes.data, None if ww is None else ww.ctypes.data)

    def inject_depth_table(self, table):
        """Test hook: BayesCalc::ln_pmf values [101][width] of the GPU's extended depth table for depths beyond the 256 of the LinearCache."""
        t = np.ascontiguousarray(table, dtype=np.float64)
        assert t.ndim == 2 and t.shape[0] == cdefs.GC_BINS
        lib().orc_locus_inject_depth_table(self._h, t.shape[1], t.ctypes.data)

    def load(self, chunk):
        """AllAlignments::load -> OracleAlns (raises ValueError with the error code on invalid data)."""
        err = C.c_int(0)
        hs = chunk.host_struct()
        h = lib().orc_load(self._h, C.byref(hs), C.byref(err))
        if not h:
            raise ValueError(f"orc_load failed: {err.value}")
        return OracleAlns(h, self.n_alleles)

    def load_mt(self, chunk, threads, hap=None):
        """The same load with the reference's thread structure (serial BAM loop + `threads` grouping workers).
        Returns (OracleAlns, (serial seconds, grouping seconds))."""
        err = C.c_int(0)
        hs = chunk.host_struct()
        secs = np.zeros(2)
        h = lib().orc_load_mt(self._h, C.byref(hs), None if hap is None else hap._h, threads, secs.ctypes.data, C.byref(err))
        if not h:
            raise ValueError(f"orc_load_mt failed: {err.value}")
        return OracleAlns(h, self.n_alleles), (float(secs[0]), float(secs[1]))

    def load_recover(self, chunk, hap):
        """AllAlignments::load with alignment recovery (opt_hap_alns = Some)."""
        err = C.c_int(0)
        hs = chunk.host_struct()
        h = lib().orc_load_recover(self._h, C.byref(hs), hap._h, C.byref(err))
        if not h:
            raise ValueError(f"orc_load_recover failed: {err.value}")
        return OracleAlns(h, self.n_alleles)


OPC = {"M": 0, "I": 1, "D": 2, "S": 4, "H": 5, "=": 7, "X": 8}
OPS = {v: k for k, v in OPC.items()}


def cigar_words(s):
    """'10=1X3I' -> raw BAM words"""
    out, num = [], ""
    for ch in s:
        if ch.isdigit():
            num += ch
        else:
            out.append((int(num) << 4) | OPC[ch]); num = ""
    return np.array(out, dtype=np.uint32)


def cigar_str(words):
    return "".join(f"{int(w) >> 4}{OPS[int(w) & 15]}" for w in words)


def dp_align(s1, s2, match_bonus=0, mode=0):
    """The oracle's aligner on its own (reference s1, query s2): (penalty, cigar string)."""
    a = np.frombuffer(s1, dtype=np.uint8); b = np.frombuffer(s2, dtype=np.uint8)
    out = np.zeros(len(a) + len(b) + 2, dtype=np.uint32)
    n = U32()
    pen = lib().orc_dp_align(a.ctypes.data, len(a), b.ctypes.data, len(b), match_bonus, mode, out.ctypes.data, len(out), C.byref(n))
    return pen, cigar_str(out[:n.value])


class HapAlns:
    """HapAlns (seq/transfer.rs:21-67) from pairwise alignments of the alleles."""

    def __init__(self, n_contigs, transfer_fails=3, max_div=0.05):
        self._h = lib().orc_hap_alns_new(n_contigs, transfer_fails, max_div)
        self.entries = []

    def add(self, id1, id2, cigar):
        w = cigar_words(cigar) if isinstance(cigar, str) else np.ascontiguousarray(cigar, dtype=np.uint32)
        nm = int(sum(int(x) >> 4 for x in w if int(x) & 15 == 7))
        ln = int(sum(int(x) >> 4 for x in w))
        lib().orc_hap_alns_add(self._h, id1, id2, w.ctypes.data, len(w), nm, ln)
        self.entries.append((id1, id2, w, nm, ln))

    def sort(self):
        lib().orc_hap_alns_sort(self._h)

    def transfer_one(self, source, target, start, read_cigar, read_seq, target_seq):
        rc = cigar_words(read_cigar)
        rs = np.frombuffer(read_seq, dtype=np.uint8); ts = np.frombuffer(target_seq, dtype=np.uint8)
        out = np.zeros(2 * len(rs) + 16, dtype=np.uint32)
        n = U32()
        st = lib().orc_transfer_one(self._h, source, target, start, rc.ctypes.data, len(rc), rs.ctypes.data, len(rs), ts.ctypes.data, len(ts),
                                    out.ctypes.data, len(out), C.byref(n))
        return int(st), cigar_str(out[:n.value])

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_hap_alns_free(self._h); self._h = None


class OracleAlns:
    def __init__(self, h, n_alleles):
        self._h = h
        self.n_alleles = n_alleles
        L = lib()
        self.n_pairs = int(L.orc_alns_n_pairs(h))
        self.n_good = int(L.orc_alns_n_good(h))
        self.status = np.zeros(self.n_pairs, dtype=np.uint8)
        self.weight = np.zeros(self.n_pairs, dtype=np.float64)
        self.unmapped_prob = np.zeros(self.n_pairs, dtype=np.float64)
        self.uniq_kmers = np.zeros(2 * self.n_pairs, dtype=np.uint16)
        L.orc_alns_status(h, self.status.ctypes.data, self.weight.ctypes.data, self.unmapped_prob.ctypes.data,
                          self.uniq_kmers.ctypes.data)
        self.pa_off = np.zeros(self.n_pairs + 1, dtype=np.uint64)
        n = int(L.orc_alns_pair_alns(h, self.pa_off.ctypes.data, None, 0))
        self.pair_alns = np.zeros(n, dtype=PAIR_ALN_DTYPE)
        L.orc_alns_pair_alns(h, None, self.pair_alns.ctypes.data, n)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_alns_free(self._h)
            self._h = None

    def best_aln_matrix(self):
        out = np.zeros((self.n_alleles, self.n_good), dtype=np.float64)
        lib().orc_best_aln_matrix(self._h, out.ctypes.data)
        return out


def alns_from_arrays(n_alleles, status, weight, unmapped_prob, pa_off, pair_alns):
    """Test hook: OracleAlns over externally produced AllAlignments products (e.g. the GPU's)."""
    status = np.ascontiguousarray(status, dtype=np.uint8)
    weight = np.ascontiguousarray(weight, dtype=np.float64)
    unm = np.ascontiguousarray(unmapped_prob, dtype=np.float64)
    off = np.ascontiguousarray(pa_off, dtype=np.uint64)
    pa = np.ascontiguousarray(pair_alns, dtype=PAIR_ALN_DTYPE)
    h = lib().orc_alns_from_arrays(len(status), n_alleles, status.ctypes.data, weight.ctypes.data, unm.ctypes.data,
                                   off.ctypes.data, pa.ctypes.data)
    return OracleAlns(h, n_alleles)


def generate_genotypes(n_alleles, ploidy):
    n = int(lib().orc_count_genotypes(n_alleles, ploidy))
    out = np.zeros((n, ploidy), dtype=np.uint16)
    w = lib().orc_generate_genotypes(n_alleles, ploidy, out.ctypes.data)
    assert w == n
    return out


def run_filter(matrix, genotypes, priors=None):
    matrix = np.ascontiguousarray(matrix, dtype=np.float64)
    genotypes = np.ascontiguousarray(genotypes, dtype=np.uint16)
    n_gt, ploidy = genotypes.shape
    scores = np.zeros(n_gt, dtype=np.float64)
    pri = None if priors is None else np.ascontiguousarray(priors, dtype=np.float64)
    lib().orc_run_filter(matrix.ctypes.data, matrix.shape[0], matrix.shape[1], genotypes.ctypes.data, n_gt, ploidy,
                         None if pri is None else pri.ctypes.data, scores.ctypes.data)
    return scores


def truncate(scores, ixs, filt_diff, min_size, threads):
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    ixs = np.ascontiguousarray(ixs, dtype=np.uint64).copy()
    m = lib().orc_truncate(scores.ctypes.data, ixs.ctypes.data, len(ixs), filt_diff, min_size, threads)
    return ixs[:int(m)]


# ---------------------------------------------------------------- solver stages
def default_solver(kind):
    s = Solver()
    lib().orc_solver_default(C.byref(s), kind)
    return s


def rng_from_seed(seed):
    r = Rng()
    lib().orc_rng_seed(C.byref(r), seed)
    return r


class OracleGtAlns:
    """GenotypeAlignments::new (assgn.rs:41-84) for genotype `ids` over the GOOD reads of `alns`."""

    def __init__(self, locus, alns, ids):
        self.locus, self.src = locus, alns
        self.ids = np.ascontiguousarray(ids, dtype=np.uint16)
        self._h = lib().orc_gt_alns_new(locus._h, alns._h, self.ids.ctypes.data, len(self.ids))
        if not self._h:
            raise ValueError("orc_gt_alns_new failed")
        L = lib()
        self.n_reads = int(L.orc_gt_alns_n_reads(self._h))
        self.n_alns = int(L.orc_gt_alns_n_alns(self._h))
        self.n_windows = int(L.orc_gt_alns_n_windows(self._h))
        self.n_nontrivial = int(L.orc_gt_alns_n_nontrivial(self._h))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_gt_alns_free(self._h)
            self._h = None

    def arrays(self):
        read_ixs = np.zeros(self.n_reads + 1, dtype=np.uint64)
        ln_prob = np.zeros(self.n_alns, dtype=np.float64)
        contig_ix = np.zeros(self.n_alns, dtype=np.uint8)
        mid1 = np.zeros(self.n_alns, dtype=np.uint32)
        mid2 = np.zeros(self.n_alns, dtype=np.uint32)
        windows = np.zeros((self.n_alns, 2), dtype=np.uint32)
        nt = np.zeros(self.n_nontrivial, dtype=np.uint64)
        lib().orc_gt_alns_get(self._h, read_ixs.ctypes.data, ln_prob.ctypes.data, contig_ix.ctypes.data,
                              mid1.ctypes.data, mid2.ctypes.data, windows.ctypes.data, nt.ctypes.data)
        return dict(read_ixs=read_ixs, ln_prob=ln_prob, contig_ix=contig_ix, mid1=mid1, mid2=mid2, windows=windows,
                    non_trivial=nt)

    def apply_tweak(self, key):
        lib().orc_gt_alns_apply_tweak(self._h, key)

    def window_distr(self):
        gc = np.zeros(self.n_windows, dtype=np.uint8)
        w = np.zeros(self.n_windows, dtype=np.float64)
        lib().orc_gt_alns_window_distr(self._h, gc.ctypes.data, w.ctypes.data)
        return gc, w

    def max_aln_lik(self):
        return lib().orc_gt_alns_max_aln_lik(self._h)

    def solve(self, solver, seed):
        """(likelihood, assignment[n_reads], (aln_lik, depth_lik)) with the solver rng = seed_from_u64(seed)."""
        assgn = np.zeros(self.n_reads, dtype=np.uint16)
        parts = np.zeros(2, dtype=np.float64)
        lik = lib().orc_solve(self._h, C.byref(solver), seed, assgn.ctypes.data, parts.ctypes.data)
        return lik, assgn, parts

    def likelihood(self, assgn):
        assgn = np.ascontiguousarray(assgn, dtype=np.uint16)
        parts = np.zeros(2, dtype=np.float64)
        lik = lib().orc_assignment_likelihood(self._h, assgn.ctypes.data, parts.ctypes.data)
        return lik, parts


def solve_stage(locus, alns, genotypes, solver, attempts, chain_seeds, priors=None, threads=None):
    """threads = None: orc_solve_stage; a number: orc_solve_stage_mt (the reference's worker threads)."""
    genotypes = np.ascontiguousarray(genotypes, dtype=np.uint16)
    n, ploidy = genotypes.shape
    seeds = np.ascontiguousarray(chain_seeds, dtype=np.uint64)
    assert len(seeds) == n * attempts
    mean = np.zeros(n)
    var = np.zeros(n)
    liks = np.zeros((n, attempts))
    pri = None if priors is None else np.ascontiguousarray(priors, dtype=np.float64)
    args = (locus._h, alns._h, genotypes.ctypes.data, n, ploidy, None if pri is None else pri.ctypes.data,
            C.byref(solver), attempts, seeds.ctypes.data, mean.ctypes.data, var.ctypes.data, liks.ctypes.data)
    if threads is None:
        lib().orc_solve_stage(*args)
    else:
        lib().orc_solve_stage_mt(*args, threads)
    return mean, var, liks


def assignment_counts(locus, alns, genotype, solver, attempts, chain_seeds):
    genotype = np.ascontiguousarray(genotype, dtype=np.uint16).reshape(-1)
    seeds = np.ascontiguousarray(chain_seeds, dtype=np.uint64)
    assert len(seeds) == attempts
    off = np.zeros(alns.n_good + 1, dtype=np.uint64)
    args = (locus._h, alns._h, genotype.ctypes.data, len(genotype), C.byref(solver), attempts, seeds.ctypes.data)
    n = lib().orc_assignment_counts(*args, off.ctypes.data, None)
    counts = np.zeros(int(n), dtype=np.uint16)
    lib().orc_assignment_counts(*args, off.ctypes.data, counts.ctypes.data)
    return off, counts


def call_checks(genotypes, ln_probs, n_reads, dist=None):
    genotypes = np.ascontiguousarray(genotypes, dtype=np.uint16)
    n, ploidy = genotypes.shape
    lp = np.ascontiguousarray(ln_probs, dtype=np.float64)
    dm = None if dist is None else np.ascontiguousarray(dist, dtype=np.uint32)
    out = np.zeros(n, dtype=np.uint32)
    wd, warn = D(), U32()
    lib().orc_call_checks(genotypes.ctypes.data, n, ploidy, lp.ctypes.data, n_reads, None if dm is None else dm.ctypes.data,
                          0 if dm is None else dm.shape[0], out.ctypes.data, C.byref(wd), C.byref(warn))
    return out, float(wd.value), int(warn.value)


def discard_improbable(lik_mean, lik_var, attempts, ixs, prob_thresh, out_size, threads):
    lik_mean = np.ascontiguousarray(lik_mean, dtype=np.float64)
    lik_var = np.ascontiguousarray(lik_var, dtype=np.float64)
    attempts = np.ascontiguousarray(attempts, dtype=np.uint32)
    ixs = np.ascontiguousarray(ixs, dtype=np.uint64).copy()
    m = lib().orc_discard_improbable(lik_mean.ctypes.data, lik_var.ctypes.data, attempts.ctypes.data, ixs.ctypes.data,
                                     len(ixs), prob_thresh, out_size, threads)
    return ixs[:int(m)]


def produce_result(lik_mean, lik_var, attempts, ixs, prob_thresh, out_bams=0):
    lik_mean = np.ascontiguousarray(lik_mean, dtype=np.float64)
    lik_var = np.ascontiguousarray(lik_var, dtype=np.float64)
    attempts = np.ascontiguousarray(attempts, dtype=np.uint32)
    ixs = np.ascontiguousarray(ixs, dtype=np.uint64)
    out_ixs = np.zeros(50, dtype=np.uint64)
    out_lp = np.zeros(50, dtype=np.float64)
    q = D()
    n = lib().orc_produce_result(lik_mean.ctypes.data, lik_var.ctypes.data, attempts.ctypes.data, ixs.ctypes.data, len(ixs),
                                 prob_thresh, out_bams, out_ixs.ctypes.data, out_lp.ctypes.data, C.byref(q))
    return out_ixs[:int(n)], out_lp[:int(n)], q.value


def canon_minimizers(seq, k, w):
    """[(pos, hash, forward)] of kmers.rs:265-331."""
    a = np.frombuffer(bytes(seq), dtype=np.uint8)
    cap = len(a) + 1
    pos = np.zeros(cap, dtype=np.uint32); hs = np.zeros(cap, dtype=np.uint64); fw = np.zeros(cap, dtype=np.uint8)
    n = lib().orc_canon_minimizers(a.ctypes.data, len(a), k, w, pos.ctypes.data, hs.ctypes.data, fw.ctypes.data, cap)
    return [(int(pos[i]), int(hs[i]), bool(fw[i])) for i in range(n)]


def fraction_approximate_u16(x):
    a, b = C.c_uint16(), C.c_uint16()
    lib().orc_fraction_approximate_u16(x, C.byref(a), C.byref(b))
    return a.value, b.value


class OracleTargets:
    """recruit::Targets (TargetBuilder + Params) of the oracle."""

    def __init__(self, k=15, w=10, match_frac=0.5, match_length=2000, thresh_kmer_count=50):
        self._h = lib().orc_targets_new(k, w, match_frac, match_length, thresh_kmer_count)
        self.n_loci = 0

    def params(self):
        a, b, c, d = C.c_uint16(), C.c_uint16(), U32(), U32()
        lib().orc_targets_params(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return (a.value, b.value), c.value, d.value

    def add_locus(self, seqs, seq_off, counts, cnt_off, base_k):
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8); seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
        counts = np.ascontiguousarray(counts, dtype=np.uint16); cnt_off = np.ascontiguousarray(cnt_off, dtype=np.uint64)
        self.n_loci += 1
        return lib().orc_targets_add_locus(self._h, len(seq_off) - 1, seqs.ctypes.data, seq_off.ctypes.data, counts.ctypes.data, cnt_off.ctypes.data, base_k)

    def finalize(self):
        lib().orc_targets_finalize(self._h)

    def entries(self):
        out = []
        m, l, d, r = U64(), U32(), C.c_uint8(), C.c_uint8()
        for i in range(lib().orc_targets_n_entries(self._h)):
            lib().orc_targets_entry(self._h, i, C.byref(m), C.byref(l), C.byref(d), C.byref(r))
            out.append((m.value, l.value, d.value, bool(r.value)))
        return out

    def recruit(self, seq1, seq2=None):
        a = np.frombuffer(bytes(seq1), dtype=np.uint8)
        b = np.frombuffer(bytes(seq2), dtype=np.uint8) if seq2 is not None else None
        out = np.zeros(max(self.n_loci, 1), dtype=np.uint32)
        n = lib().orc_recruit(self._h, a.ctypes.data, len(a), b.ctypes.data if b is not None else None, len(b) if b is not None else 0,
                              out.ctypes.data, len(out))
        return [int(x) for x in out[:n]]

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_targets_free(self._h); self._h = None
