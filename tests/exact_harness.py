"""TEST INFRASTRUCTURE. The library's exact solver (lcty_exact.cpp, host code) on a model built from the ORACLE's GenotypeAlignments —
no device involved: tests/native/exact_harness.cpp + locityper_amd/csrc/lcty_exact.cpp compiled by g++ into tests/native/_build/."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        out = os.path.join(ROOT, "tests", "native", "_build")
        os.makedirs(out, exist_ok=True)
        so = os.path.join(out, "libexact_harness.so")
        srcs = [os.path.join(ROOT, "tests", "native", "exact_harness.cpp"), os.path.join(ROOT, "locityper_amd", "csrc", "lcty_exact.cpp")]
        hdr = os.path.join(ROOT, "locityper_amd", "csrc", "lcty_exact.hpp")
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs + [hdr]):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so] + srcs)
        _LIB = C.CDLL(so)
        _LIB.exact_harness_depth_needed.restype = C.c_uint64
        _LIB.exact_harness_solve.restype = C.c_int
    return _LIB


class Model:
    """exact::Model of one (genotype, attempt) from the oracle's arrays (tests/oracle_ffi.py: OracleGtAlns.arrays() after apply_tweak,
    window_distr()): the non-trivial reads with their locations in the oracle's order (best first, windows.rs:762-797)."""

    def __init__(self, arrays, gc, weight, n_alleles_windows=None):
        rix = arrays["read_ixs"].astype(np.int64)
        lp, win = arrays["ln_prob"], arrays["windows"]
        nloc = np.diff(rix)
        self.reads = np.nonzero(nloc >= 2)[0]                        # index of every non-trivial read among the genotype's reads
        self.n_reads = len(nloc)
        first = [0]
        sel = []
        for r in self.reads:
            sel.extend(range(rix[r], rix[r + 1])); first.append(len(sel))
        sel = np.asarray(sel, dtype=np.int64)
        self.first = np.asarray(first, dtype=np.uint32)
        self.lp = np.ascontiguousarray(lp[sel], dtype=np.float64)
        self.wa = np.ascontiguousarray(win[sel, 0], dtype=np.uint32)
        self.wb = np.ascontiguousarray(win[sel, 1], dtype=np.uint32)
        self.ww = np.ascontiguousarray(weight, dtype=np.float64)
        self.gcb = np.ascontiguousarray(gc, dtype=np.uint8)
        tw = len(gc)
        start = rix[:-1]                                             # every read at its location 0
        self.depth0 = np.bincount(win[start].reshape(-1).astype(np.int64), minlength=tw).astype(np.uint32)
        self.aln0 = float(np.sum(lp[start]))
        self.afw = np.asarray(n_alleles_windows if n_alleles_windows is not None else [0, tw], dtype=np.uint32)
        self.tw = tw

    def _args(self):
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        return (C.c_uint32(len(self.reads)), C.c_uint32(self.tw), p(self.first), p(self.lp), p(self.wa), p(self.wb), p(self.ww), p(self.gcb), p(self.depth0))

    def depth_needed(self):
        return int(lib().exact_harness_depth_needed(*self._args(), self.afw.ctypes.data_as(C.c_void_p), C.c_uint32(len(self.afw))))

    def solve(self, lut, aln_contrib, depth_contrib, rel_gap=1e-4, node_limit=20_000_000, trace=0):
        """(answered, assignment over ALL reads of the genotype [uint16], value, (aln_lik, depth_lik), nodes, free reads at the root)."""
        lut = np.ascontiguousarray(lut, dtype=np.float64)
        assign = np.zeros(len(self.reads), dtype=np.uint8)
        value, parts = C.c_double(0), (C.c_double * 2)()
        nodes, n_free = C.c_uint64(0), C.c_uint32(0)
        rc = lib().exact_harness_solve(*self._args(), C.c_double(self.aln0), self.afw.ctypes.data_as(C.c_void_p), C.c_uint32(len(self.afw)),
                                       C.c_double(aln_contrib), C.c_double(depth_contrib), C.c_uint64(node_limit), C.c_double(rel_gap), C.c_int(trace),
                                       lut.ctypes.data_as(C.c_void_p), C.c_uint32(lut.shape[1]), C.c_uint32(lut.shape[0]),
                                       assign.ctypes.data_as(C.c_void_p), C.byref(value), parts, C.byref(nodes), C.byref(n_free))
        full = np.zeros(self.n_reads, dtype=np.uint16)
        full[self.reads] = assign
        return rc == 0, full, value.value, (parts[0], parts[1]), nodes.value, n_free.value
