"""The host-side entry points of the library (lcty_kmer_counts_parse, lcty_call_checks, lcty_counts_to_posteriors, lcty_truncate, ...)
once more on the GPU box: tests/test_host_api.py needs no device and runs in the CPU tier; re-exported here under the `gpu` marker so
that the driver's `-m gpu` run observes them next to the kernels (same functions, no second copy)."""
import pytest

from tests import test_host_api as _H

pytestmark = pytest.mark.gpu

for _name in dir(_H):
    if _name.startswith("test_") and _name != "test_compute_entry_points_fail_loudly_without_gpu":
        globals()[_name + "_on_the_gpu_box"] = getattr(_H, _name)
del _name


def test_chunks_in_page_locked_memory(gpu_ctx):
    """lcty_host_alloc: a chunk whose arrays lie in page-locked memory gives what the same chunk in ordinary memory gives (the
    copies take the direct path over PCIe); the memory goes back with the last view of it."""
    import gc
    import numpy as np
    from locityper_amd import api, synth
    L = synth.SynthLocus(8, 3000, seed=12, base_len=12000)
    p = api.resolve_params(api.default_params(), L.bg)
    loc = api.Locus(gpu_ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    ch = L.reads(0, 3000)
    a = api.AllAlignments.load(loc, ch)
    pinned = gpu_ctx.pinned_chunk(ch)
    assert np.array_equal(pinned.cigar, ch.cigar) and np.array_equal(pinned.recs, ch.recs)
    b = api.AllAlignments.load(loc, pinned)
    for x, y in zip(a.status(), b.status()):
        assert np.array_equal(x, y)
    assert np.array_equal(a.best_aln_matrix(), b.best_aln_matrix())
    del pinned, b
    gc.collect()
    big = gpu_ctx.pinned_like(np.arange(1 << 20, dtype=np.uint32))
    assert int(big[-1]) == (1 << 20) - 1
