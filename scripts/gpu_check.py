"""Developer check: HIP path vs oracle on synthetic data (run on the GPU box through gpurun)."""
import sys
import time
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locityper_amd import api, synth, cdefs  # noqa: E402
from tests import oracle_ffi as O  # noqa: E402


def compare(n_alleles, n_pairs, n_expected, tech=cdefs.TECH_ILLUMINA, read_len=150):
    print(f"=== A={n_alleles} pairs={n_pairs} tech={tech}")
    L = synth.SynthLocus(n_alleles, n_expected, technology=tech, read_len=read_len)
    ch = L.reads(0, n_pairs)
    p = api.default_params()
    api.resolve_params(p, L.bg)
    ctx = api.Context(0)
    t = time.time()
    loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    print("locus create", time.time() - t)
    ol = O.OracleLocus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    print("unique kmers", loc.n_unique_kmers(), ol.n_unique_kmers())
    assert loc.n_unique_kmers() == ol.n_unique_kmers()
    for a in (0, n_alleles - 1):
        g = loc.contig_info(a)
        o = ol.contig_info(a)
        for x, y in zip(g, o):
            assert np.array_equal(np.asarray(x), np.asarray(y)), "contig_info mismatch"
    t = time.time()
    aa = api.AllAlignments.load(loc, ch)
    ctx.synchronize()
    print("gpu load", time.time() - t)
    t = time.time()
    oa = ol.load(ch)
    print("oracle load", time.time() - t)
    st, w, unm, uk = aa.status()
    print("status gpu", np.bincount(st, minlength=4), "oracle", np.bincount(oa.status, minlength=4))
    bad = np.nonzero(st != oa.status)[0]
    print("status mismatches", len(bad), bad[:10])
    assert len(bad) == 0
    assert np.array_equal(uk, oa.uniq_kmers), (np.nonzero(uk != oa.uniq_kmers)[0][:10])
    print("max |weight diff|", np.abs(w - oa.weight).max(), "max |unm diff|", np.abs(unm - oa.unmapped_prob).max())
    assert np.allclose(w, oa.weight, rtol=0, atol=1e-12)
    assert np.allclose(unm, oa.unmapped_prob, rtol=0, atol=1e-9)
    M = aa.best_aln_matrix()
    Mo = oa.best_aln_matrix()
    print("matrix", M.shape, "max abs diff", np.abs(M - Mo).max(), "bit-equal", np.array_equal(M, Mo))
    assert np.abs(M - Mo).max() < 1e-9
    off, pa = aa.pair_alns()
    assert np.array_equal(off, oa.pa_off), "pair-aln offsets differ"
    for f in ("contig", "ix1", "ix2", "mid1", "mid2"):
        assert np.array_equal(pa[f], oa.pair_alns[f]), f
    print("pair alns", len(pa), "max lp diff", np.abs(pa["ln_prob"] - oa.pair_alns["ln_prob"]).max())
    t = time.time()
    sc = aa.run_filter()
    print("gpu filter", time.time() - t)
    gts = O.generate_genotypes(n_alleles, 2)
    assert np.array_equal(gts, api.generate_genotypes(n_alleles, 2))
    t = time.time()
    so = O.run_filter(Mo, gts)
    print("oracle filter", time.time() - t)
    rel = np.abs(sc - so).max() / np.abs(so).max()
    print("scores rel diff", rel, "argmax", gts[np.argmax(sc)], gts[np.argmax(so)], "true", L.true_genotype)
    assert rel < 1e-12
    # generic path with explicit genotypes + priors
    sub = gts[::7]
    pri = -np.arange(len(sub), dtype=np.float64)
    sg = aa.run_filter(sub, pri)
    assert np.allclose(sg, so[::7] + pri, rtol=1e-12)
    k1 = api.truncate_ixs(sc, np.arange(len(sc)), p.filt_diff, 50, 8)
    k2 = O.truncate(so, np.arange(len(so)), p.filt_diff, 50, 8)
    print("truncate", len(k1), len(k2))
    assert set(k1.tolist()) == set(k2.tolist())
    print("OK")


if __name__ == "__main__":
    print("devices:", api.device_count())
    compare(8, 2000, 10000)
    compare(256, 1024, 1_000_000)
    compare(20, 300, 3000, tech=cdefs.TECH_NANOPORE, read_len=3000)
