"""The C ABI from compiled code: examples/genotype_locus.cpp is built against the header and the two shared
libraries and run on the GPU box (no Python between the program and the library)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_example(out, source="genotype_locus.cpp", threads=False):
    lib_dir = os.path.join(ROOT, "locityper_amd")
    synth_dir = os.path.join(lib_dir, "synth")
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", source),
           "-o", out, "-L" + lib_dir, "-llocityper_hip", "-L" + synth_dir, "-llcty_synth",
           "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + synth_dir, "-Wl,-rpath,/opt/rocm/lib"] + (["-pthread"] if threads else [])
    subprocess.run(cmd, check=True, capture_output=True, text=True)


def test_example_compiles_against_the_header(tmp_path):
    build_example(str(tmp_path / "genotype_locus"))
    build_example(str(tmp_path / "genotype_dir"), "genotype_dir.cpp")
    build_example(str(tmp_path / "solver_trait_twin"), "solver_trait_twin.cpp", threads=True)


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


@pytest.mark.gpu
def test_the_shims_call_sequence_from_compiled_code_on_worker_threads(tmp_path):
    """shim/src/solvers/hip.rs cannot be compiled here (no rustc); examples/solver_trait_twin.cpp is `HipSolver::solve_nontrivial`
    statement for statement in C++: flatten the object, lcty_gt_alns_deepest, rows of ln_pmf, lcty_rng_seed_from_u64 of one draw of the
    caller's generator, lcty_solve_given_tables — called from four worker threads that share one context, as solve_multi_thread calls
    `stage.solver.solve(&gt_alns, rng)` (solve.rs:1010-1017, 1124-1125). Twelve oracle-built GenotypeAlignments (three solver
    settings, heterozygous / homozygous / ploidy 3): the program's assignments must be the oracle's chains read for read."""
    import struct
    import numpy as np
    from locityper_amd import api, cdefs, synth
    from tests import oracle_ffi as O
    L = synth.SynthLocus(8, 3000, seed=31, base_len=20000)
    p = api.resolve_params(api.default_params(), L.bg)
    ctx = api.Context(0)
    loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    ol = O.OracleLocus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    oa = ol.load(L.reads(0, 3000))
    width = 1 << 13
    table = loc.depth_table(width)                       # the rows a LinearCache would give: one per GC bin
    ol.inject_tables(loc.depth_lut(), None)
    ol.inject_depth_table(table)
    g0 = api.default_solver(cdefs.SOLVER_GREEDY)
    g1 = api.default_solver(cdefs.SOLVER_GREEDY); g1.best_start, g1.sample_size, g1.plato_size = 0, 4, 40
    a0 = api.default_solver(cdefs.SOLVER_ANNEAL); a0.anneal_steps, a0.plato_size = 3000, 1500
    objects, want = [], []
    for k, (ids, sv) in enumerate([(ids, sv) for ids in [(0, 1), (3, 3), (2, 5, 7), tuple(int(x) for x in L.true_genotype)] for sv in (g0, g1, a0)]):
        g = O.OracleGtAlns(ol, oa, ids)
        g.apply_tweak(777 + k)
        a = g.arrays()
        gc, w = g.window_distr()
        draw = 1000 + 17 * k
        state = api.rng_seed_from_u64(draw)
        seed = api.rng_next_u64(state)                   # what lcty_solve_given_tables takes from the four words the twin seeds with `draw`
        want.append(g.solve(sv, seed))
        wsh = np.array([2, len(w)], dtype=np.uint32)       # GenotypeWindows::wshifts as one run: only the exact solver's search order looks at it
        objects.append((a, np.where(w != 0.0, gc.astype(np.uint32), 0xFFFFFFFF).astype(np.uint32), w, wsh, sv, draw))
    dump = tmp_path / "objects.bin"
    with open(dump, "wb") as f:
        f.write(struct.pack("<III", table.shape[0], width, len(objects)))
        f.write(np.ascontiguousarray(table, dtype=np.float64).tobytes())
        for a, distr, w, wsh, sv, draw in objects:
            n_reads = len(a["read_ixs"]) - 1
            f.write(struct.pack("<QII", n_reads, len(w), len(wsh) - 1))
            f.write(a["read_ixs"].astype(np.uint64).tobytes())
            f.write(a["ln_prob"].astype(np.float64).tobytes())           # read by read: the CSR order is the per-read order
            f.write(np.ascontiguousarray(a["windows"], dtype=np.uint32).tobytes())
            f.write(distr.tobytes()); f.write(np.ascontiguousarray(w, dtype=np.float64).tobytes()); f.write(wsh.tobytes())
            f.write(struct.pack("<dd", 1.0 + p.lik_skew, 1.0 - p.lik_skew))
            f.write(bytes(sv)); f.write(struct.pack("<Q", draw))
    exe = str(tmp_path / "solver_trait_twin")
    build_example(exe, "solver_trait_twin.cpp", threads=True)
    out = tmp_path / "assignments.bin"
    r = subprocess.run([exe, str(dump), str(out), "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(out, "rb").read()
    at = 0
    for (olik, oassgn, _), (a, *_rest) in zip(want, objects):
        n = struct.unpack_from("<Q", raw, at)[0]; at += 8
        assgn = np.frombuffer(raw, dtype=np.uint16, count=n, offset=at); at += 2 * n
        lik = struct.unpack_from("<d", raw, at)[0]; at += 8
        assert n == len(oassgn) and np.array_equal(assgn, oassgn) and abs(lik - olik) <= 1e-9 * abs(olik)
    assert at == len(raw)
