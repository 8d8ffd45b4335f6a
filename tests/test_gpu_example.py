"""The C ABI from compiled code: examples/genotype_locus.cpp is built against the header and the two shared
libraries and run on the GPU box (no Python between the program and the library)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_example(out):
    lib_dir = os.path.join(ROOT, "locityper_amd")
    synth_dir = os.path.join(lib_dir, "synth")
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "genotype_locus.cpp"),
           "-o", out, "-L" + lib_dir, "-llocityper_hip", "-L" + synth_dir, "-llcty_synth",
           "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + synth_dir, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)


def test_example_compiles_against_the_header(tmp_path):
    build_example(str(tmp_path / "genotype_locus"))


@pytest.mark.gpu
def test_example_genotypes_config1(tmp_path):
    exe = str(tmp_path / "genotype_locus")
    build_example(exe)
    r = subprocess.run([exe, "8", "10000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("called ") and " warnings 0 " in r.stdout
