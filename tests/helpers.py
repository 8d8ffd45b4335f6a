"""Shared builders for tests: small loci, hand-written read chunks, comparison helpers."""
import ctypes as C

import numpy as np

from locityper_amd import cdefs, synth
from locityper_amd.cdefs import ReadsChunk
from tests import oracle_ffi as O
from tests import pyref

F = cdefs  # flags / ops


def make_bg(technology=cdefs.TECH_ILLUMINA, paired=True, window=100, neighb=300):
    import math
    bg = cdefs.Bg()
    px, pi, pd = 0.003, 0.001, 0.001
    vals = [math.log(1 - px - pi - pd), math.log(px), math.log(pi), math.log(pd), math.log(max(px, pi))]
    for i, v in enumerate(vals):
        bg.op_lnprobs[i] = v
    bg.edit_alpha, bg.edit_beta = 0.6, 90.0
    bg.is_paired = int(paired)
    bg.ins_n, bg.ins_p = 450.0 * 450.0 / (6400.0 - 450.0), 450.0 / 6400.0
    for i in range(cdefs.GC_BINS):
        bg.depth_n[i], bg.depth_p[i] = 20.0, 2.0 / 3.0
    bg.window, bg.neighb = window, neighb
    bg.technology = technology
    if technology == cdefs.TECH_ILLUMINA:
        bg.edit_kind, bg.edit_p1, bg.edit_p2 = cdefs.EDIT_FRACTION, 0.03, 0.06
    else:
        bg.edit_kind, bg.edit_p1, bg.edit_p2 = cdefs.EDIT_PVALUE, 0.99, 0.999
    return bg


def random_alleles(n, length, seed=1, snp_rate=0.01):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 4, length)
    out = []
    for _ in range(n):
        a = base.copy()
        m = rng.random(length) < snp_rate
        a[m] = (a[m] + rng.integers(1, 4, m.sum())) % 4
        out.append(bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[a]))
    return out


def locus_arrays(alleles, k, counts=None):
    seqs = np.frombuffer(b"".join(alleles), dtype=np.uint8).copy()
    seq_off = np.zeros(len(alleles) + 1, dtype=np.uint64)
    np.cumsum([len(a) for a in alleles], out=seq_off[1:])
    if counts is None:
        counts = [np.zeros(len(a) + 1 - k, dtype=np.uint16) for a in alleles]
    cnt_off = np.zeros(len(alleles) + 1, dtype=np.uint64)
    np.cumsum([len(c) for c in counts], out=cnt_off[1:])
    return seqs, seq_off, np.concatenate(counts).astype(np.uint16), cnt_off, counts


def oracle_and_pyref(alleles, k, bg, params, counts=None):
    seqs, seq_off, cflat, cnt_off, counts = locus_arrays(alleles, k, counts)
    ol = O.OracleLocus(seqs, seq_off, cflat, cnt_off, k, bg, params)

    def edit_thr(rl):
        g, p = C.c_uint32(), C.c_uint32()
        O.lib().orc_edit_thresholds(C.byref(bg), rl, C.byref(g), C.byref(p))
        return g.value, p.value
    pl = pyref.PyLocus(alleles, counts, k, bg, params, ol.insert_lnprob if bg.is_paired else (lambda sz: 0.0),
                       ol.insert_penalty() if bg.is_paired else float("nan"), edit_thr)
    return ol, pl, (seqs, seq_off, cflat, cnt_off)


def compare_load(oa, py_res, exact=True):
    """OracleAlns vs pyref.load output."""
    assert oa.n_pairs == len(py_res)
    for r, (status, weight, unm, uks, pairs) in enumerate(py_res):
        assert oa.status[r] == status, (r, oa.status[r], status)
        assert tuple(oa.uniq_kmers[2 * r:2 * r + 2]) == uks, r
        if exact:
            assert oa.weight[r] == weight and oa.unmapped_prob[r] == unm, r
        else:
            assert abs(oa.weight[r] - weight) < 1e-12 and abs(oa.unmapped_prob[r] - unm) < 1e-9
        lo, hi = int(oa.pa_off[r]), int(oa.pa_off[r + 1])
        assert hi - lo == len(pairs), (r, hi - lo, len(pairs))
        for t, pa in enumerate(pairs):
            o = oa.pair_alns[lo + t]
            got = (float(o["ln_prob"]), int(o["contig"]), int(o["ix1"]), int(o["mid1"]), int(o["ix2"]), int(o["mid2"]))
            if exact:
                assert got == pa, (r, t, got, pa)
            else:
                assert got[1:] == pa[1:] and abs(got[0] - pa[0]) < 1e-9, (r, t, got, pa)


def compare_gpu_to_oracle(aa, oa, tol_lp=1e-9, index_fields=("ix1", "ix2")):
    """api.AllAlignments (HIP) vs OracleAlns: integers bit-exact, likelihoods within tolerance. index_fields = () leaves the
    record indices out (after alignment recovery they number the merged record table on the device side)."""
    st, w, unm, uk = aa.status()
    assert np.array_equal(st, oa.status), np.nonzero(st != oa.status)[0][:10]
    assert np.array_equal(uk, oa.uniq_kmers)          # bit-exact contract for k-mer counts
    assert np.allclose(w, oa.weight, rtol=0, atol=1e-12)
    assert np.allclose(unm, oa.unmapped_prob, rtol=0, atol=tol_lp)
    off, pa = aa.pair_alns()
    assert np.array_equal(off, oa.pa_off)
    for f in ("contig", "mid1", "mid2") + tuple(index_fields):
        assert np.array_equal(pa[f], oa.pair_alns[f]), f
    if len(pa):
        assert np.abs(pa["ln_prob"] - oa.pair_alns["ln_prob"]).max() <= tol_lp
    M, Mo = aa.best_aln_matrix(), oa.best_aln_matrix()
    assert M.shape == Mo.shape
    if M.size:
        assert np.abs(M - Mo).max() <= 1e-5       # north_star tolerance on log-likelihoods
        assert np.abs(M - Mo).max() <= tol_lp
    return M, Mo


def noisy_read(rng, hap, start, length, err=0.03):
    """A read copied from hap[start:start+length] with substitutions / insertions / deletions and its true CIGAR."""
    acgt = b"ACGT"
    seq = bytearray(); ops = []

    def push(op, n=1):
        if ops and ops[-1][0] == op: ops[-1][1] += n
        else: ops.append([op, n])
    i = start
    while i < start + length:
        r = rng.random()
        if r < err / 3:
            seq.append(int(rng.choice([c for c in acgt if c != hap[i]]))); push("X"); i += 1
        elif r < 2 * err / 3 and ops:
            k = int(rng.integers(1, 4)); seq.extend(rng.choice(list(acgt), k).tolist()); push("I", k)
        elif r < err and ops and ops[-1][0] == "=":
            k = int(rng.integers(1, 4)); push("D", k); i += k
        else:
            seq.append(hap[i]); push("="); i += 1
    while ops and ops[-1][0] in "DI":                                         # an alignment ends on an aligned base
        op, n = ops.pop()
        if op == "I": del seq[-n:]
    return bytes(seq), "".join(f"{n}{op}" for op, n in ops)
