"""The file formats at the edges of the path (lcty_io.hip: gzip / BGZF / LZ4 / brotli containers, kmers.bin, distr.gz, res.json, the
aln.bam reader) once more on the GPU box: tests/test_io_formats.py needs no device and runs in the CPU tier; re-exported here under the
`gpu` marker so that the driver's `-m gpu` run observes them next to the kernels (same functions, no second copy)."""
import pytest

from tests import test_io_formats as _F

pytestmark = pytest.mark.gpu

for _name in dir(_F):
    if _name.startswith("test_"):
        globals()[_name + "_on_the_gpu_box"] = getattr(_F, _name)
del _name
