"""The C ABI from compiled code: examples/genotype_locus.cpp is built against the header and the two shared
libraries and run on the GPU box (no Python between the program and the library)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_example(out, source="genotype_locus.cpp"):
    lib_dir = os.path.join(ROOT, "locityper_amd")
    synth_dir = os.path.join(lib_dir, "synth")
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", source),
           "-o", out, "-L" + lib_dir, "-llocityper_hip", "-L" + synth_dir, "-llcty_synth",
           "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + synth_dir, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)


def test_example_compiles_against_the_header(tmp_path):
    build_example(str(tmp_path / "genotype_locus"))
    build_example(str(tmp_path / "genotype_dir"), "genotype_dir.cpp")


@pytest.mark.gpu
def test_example_genotypes_config1(tmp_path):
    exe = str(tmp_path / "genotype_locus")
    build_example(exe)
    r = subprocess.run([exe, "8", "10000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("called ") and " warnings 0 " in r.stdout


@pytest.mark.gpu
def test_example_from_a_locityper_directory(tmp_path):
    """Files in, files out: a synthetic locus written in the layout `locityper genotype` works on (haplotypes.fa.gz, kmers.bin.lz4,
    distr.gz, aln.bam) goes through examples/genotype_dir.cpp — C ABI only — and comes back as res.json.gz, which must hold what
    extra/into_csv.py:65-100 reads and name the genotype the reads were drawn from, and as alns/00.bam + .bai."""
    import gzip
    import json
    import sys
    root = str(tmp_path / "lcty")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "make_locityper_dir.py"), root, "--alleles", "8", "--pairs", "6000",
                        "--base-len", "30000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    exe = str(tmp_path / "genotype_dir")
    build_example(exe, "genotype_dir.cpp")
    r = subprocess.run([exe, root, "L1", "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    truth = json.load(open(os.path.join(root, "truth.json")))
    res = json.load(gzip.open(os.path.join(root, "OUT", "loci", "L1", "res.json.gz"), "rt"))
    assert res["genotype"] == ",".join(truth["genotype"]) and res["quality"] > 20 and 0 < res["total_reads"] <= truth["pairs"]
    assert {"total_reads", "quality", "unexpl_reads", "genotype", "options"} <= set(res) and res["options"][0]["genotype"] == res["genotype"]
    assert abs(sum(o["prob"] for o in res["options"]) - 1.0) < 1e-9 and "warnings" not in res
    bam = os.path.join(root, "OUT", "loci", "L1", "alns", "00.bam")
    raw = gzip.open(bam, "rb").read()
    assert raw[:4] == b"BAM\x01" and os.path.getsize(bam + ".bai") > 32
    assert r.stdout.startswith("genotype " + res["genotype"])


@pytest.mark.gpu
def test_example_from_a_directory_of_a_basis_run(tmp_path):
    """The same with the layout of a run on basis haplotypes: aln.bam holds the primary alignments only, haplotypes.paf.gz the pairwise
    haplotype alignments (lcty_paf_read -> lcty_locus_set_hap_alns -> lcty_recover_alignments); the call must be the genotype the
    reads were drawn from."""
    import gzip
    import json
    import sys
    root = str(tmp_path / "lcty")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "make_locityper_dir.py"), root, "--alleles", "6", "--pairs", "5000",
                        "--base-len", "20000", "--paf"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert os.path.getsize(os.path.join(root, "DB", "loci", "L1", "haplotypes.paf.gz")) > 100
    exe = str(tmp_path / "genotype_dir")
    build_example(exe, "genotype_dir.cpp")
    r = subprocess.run([exe, root, "L1", "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    truth = json.load(open(os.path.join(root, "truth.json")))
    res = json.load(gzip.open(os.path.join(root, "OUT", "loci", "L1", "res.json.gz"), "rt"))
    assert res["genotype"] == ",".join(truth["genotype"]) and res["quality"] > 20
    assert res["dist_type"] == "edit" and res["options"][0]["dist_to_primary"] == 0 and res["weight_dist"] >= 0.0
    assert all(isinstance(o["dist_to_primary"], int) for o in res["options"])
    recovered = int(r.stdout.strip().split(" recovered ")[1])
    assert recovered > 4 * 5000                                           # every read pair reaches the other five alleles
    bam = os.path.join(root, "OUT", "loci", "L1", "alns", "00.bam")      # the placements on the call, transferred alignments included
    raw = gzip.open(bam, "rb").read()
    assert raw[:4] == b"BAM\x01" and os.path.getsize(bam + ".bai") > 32
    assert int(r.stdout.split(" bam_records ")[1].split()[0]) >= 2 * 4000
