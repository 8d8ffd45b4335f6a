"""The tables the solver-stage parity tests hand from the GPU to the oracle (tests/test_gpu_solve.py injects the device's depth
table and window weights so that every chain must follow the oracle move for move) pinned in full against the oracle's own values:
every entry of the 101 x 256 LinearCache, every entry of the extended table up to depth 8 192, every per-position window weight of
every allele, every insert-size LUT entry."""
import ctypes as C

import numpy as np
import pytest

from locityper_amd import api, cdefs, synth
from tests import oracle_ffi as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tech,rl,n_pairs", [(cdefs.TECH_ILLUMINA, 150, 200_000), (cdefs.TECH_NANOPORE, 5000, 20_000)])
def test_depth_tables_entry_by_entry(gpu_ctx, tech, rl, n_pairs):
    L = synth.SynthLocus(6, n_pairs, technology=tech, read_len=rl, base_len=20_000)
    p = api.resolve_params(api.default_params(), L.bg)
    loc = api.Locus(gpu_ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    want = np.zeros((cdefs.GC_BINS, 8192))
    O.lib().orc_depth_table(C.byref(L.bg), C.byref(p), 0, 8192, want.ctypes.data)
    lut = loc.depth_lut()
    assert lut.shape == (cdefs.GC_BINS, 256)
    # LinearCache (host libm lgamma against the oracle's statrs Lanczos): all 25 856 entries
    assert np.all(np.isfinite(lut)) and np.abs(lut - want[:, :256]).max() <= 1e-10
    ext = loc.depth_table(8192)
    assert ext.shape == (cdefs.GC_BINS, 8192)
    assert np.array_equal(ext[:, :256], lut)                                 # the first columns ARE the LinearCache
    # device lgamma / exp / log against the oracle (statrs' Lanczos): all 827 392 entries; values reach -1e4, so relative.
    # lgamma of arguments around 1e4 is ~8e4 and the two implementations differ in its last digits: 2e-11 of the entry observed
    err = np.abs(ext - want) / np.maximum(1.0, np.abs(want))
    assert np.all(np.isfinite(ext)) and err.max() <= 1e-10, (err.max(), np.unravel_index(np.argmax(err), err.shape))
    # a wider table later leaves the narrower read-back unchanged
    ext2 = loc.depth_table(32768)
    assert np.array_equal(ext2[:, :8192], ext)


@pytest.mark.parametrize("n_alleles,tech,rl", [(12, cdefs.TECH_ILLUMINA, 150), (5, cdefs.TECH_NANOPORE, 4000)])
def test_every_window_weight_and_insert_size(gpu_ctx, n_alleles, tech, rl):
    L = synth.SynthLocus(n_alleles, 10_000, technology=tech, read_len=rl)
    p = api.resolve_params(api.default_params(), L.bg)
    loc = api.Locus(gpu_ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    ol = O.OracleLocus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    ww = loc.window_weights()
    want = np.zeros_like(ww)
    O.lib().orc_locus_window_weights(ol._h, want.ctypes.data)
    assert len(ww) == int(sum(int(L.seq_off[a + 1] - L.seq_off[a]) - L.bg.neighb + 1 for a in range(n_alleles)))
    # two powf per position on each side (device pow against libm pow): 1e-13 of the value
    assert np.abs(ww - want).max() <= 1e-13 and 0.0 <= ww.min() and ww.max() <= 1.0 and ww.std() > 0.001
    if L.bg.is_paired:
        sizes = np.arange(0, 70000, dtype=np.uint32)                       # the whole LUT and 4 000 sizes beyond it
        got, pen = loc.insert_lnprob(sizes)
        wanti = np.array([ol.insert_lnprob(int(s)) for s in sizes[::7]])
        assert np.abs(got[::7] - wanti).max() <= 1e-9 * np.abs(wanti).max()
