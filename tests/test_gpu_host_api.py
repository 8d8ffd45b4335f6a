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
