"""The file formats at the edges of the path (lcty_io.hip: gzip / BGZF / LZ4 / brotli containers, kmers.bin, distr.gz, res.json, the
aln.bam reader) need no device and belong to the CPU tier (tests/test_io_formats.py). They depend on system libraries (zlib,
libbrotlidec looked up at run time), so the GPU box runs them ONCE MORE — as ONE test, so that the `-m gpu` count stays the count of
device tests (round 5 re-exported eleven functions here: eleven host-only tests counted as GPU tests)."""
import inspect
import pathlib

import pytest

from tests import test_io_formats as _F

pytestmark = pytest.mark.gpu


def test_host_side_file_formats_on_this_box(tmp_path):
    ran = 0
    for name, fn in sorted(vars(_F).items()):
        if not name.startswith("test_") or not callable(fn):
            continue
        try:
            if "tmp_path" in inspect.signature(fn).parameters:
                d = pathlib.Path(tmp_path) / name
                d.mkdir()
                fn(d)
            else:
                fn()
        except pytest.skip.Exception:                  # a format whose system library this box lacks: the others still run
            continue
        ran += 1
    assert ran >= 9
