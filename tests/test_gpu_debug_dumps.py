"""The reference's `--debug 2` files (read_kmers.csv, read_pairs.csv, sol.csv; formats at model/locs.rs:648-665, 975-999,
solvers/solve.rs:116, 841) written from the oracle and from the HIP path by scripts/debug_dumps.py must be the same text."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_debug_dumps_of_both_sides_are_the_same_text(gpu_ctx):
    spec = importlib.util.spec_from_file_location("debug_dumps", os.path.join(ROOT, "scripts", "debug_dumps.py"))
    dd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dd)
    res = dd.dumps_of_both(gpu_ctx, 8, 3000, base_len=20_000)
    hip, orc = res["hip"], res["oracle"]
    assert hip["read_kmers.csv"] == orc["read_kmers.csv"] and hip["read_kmers.csv"].count("\n") > 2500      # integers + 2 decimals
    for name, tol in (("read_pairs.csv", 1.01e-4), ("sol.csv", 1.01e-3)):
        ok, why = dd.same_but_last_digit(hip[name], orc[name], tol)
        assert ok, (name, why)
    lines = hip["read_pairs.csv"].splitlines()
    assert lines[0] == "read_hash\tcontig\tpos1\tpos2\tlik" and lines[1].split("\t")[1].startswith("a")
    sol = hip["sol.csv"].splitlines()
    assert sol[0] == "stage\tgenotype\tscore" and sol[1].startswith("0\ta0,a0\t") and sol[-1].startswith("2\t")
    assert sum(l.startswith("0\t") for l in sol) == 36 and sum(l.startswith("1\t") for l in sol) == 12
