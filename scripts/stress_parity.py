"""Developer stress run: the parity checks of tests/ over many random shapes and seeds (scoring, prefilter + both truncates, solver
stages on injected tables, candidate generation — both routes — against its restatements). Prints one line per case; exits non-zero at the first mismatch."""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from locityper_amd import api, cdefs, synth
from tests import oracle_ffi as O
from tests.helpers import compare_gpu_to_oracle
from tests.test_gpu_map import test_mapper_on_random_inputs_equals_its_restatement as mapper_case
from tests.test_gpu_map_long import test_long_route_on_random_inputs_equals_its_restatement as long_mapper_case


def one_case(ctx, rng, i):
    tech = cdefs.TECH_ILLUMINA if rng.random() < 0.7 else cdefs.TECH_NANOPORE
    A = int(rng.integers(2, 28))
    k = int(rng.choice([15, 21, 25, 31, 33, 47]))
    rl = int(rng.choice([100, 150, 250])) if tech == cdefs.TECH_ILLUMINA else int(rng.choice([1500, 4000]))
    n = int(rng.integers(200, 4000)) if tech == cdefs.TECH_ILLUMINA else int(rng.integers(50, 400))
    base = int(rng.integers(5000, 20000)) if tech == cdefs.TECH_ILLUMINA else int(rng.integers(12000, 30000))
    seed = int(rng.integers(1, 1 << 30))
    L = synth.SynthLocus(A, n, seed=seed, base_len=base, technology=tech, read_len=rl, k=k)
    p = api.default_params()
    if rng.random() < 0.3: p.kmer_soft_thresh = int(rng.integers(2, 9))
    if rng.random() < 0.3: p.min_weight = float(rng.choice([0.0, 0.01, 0.2]))
    api.resolve_params(p, L.bg)
    loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    ol = O.OracleLocus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    assert loc.n_unique_kmers() == ol.n_unique_kmers()
    ch = L.reads(0, n)
    aa, oa = api.AllAlignments.load(loc, ch), ol.load(ch)
    M, Mo = compare_gpu_to_oracle(aa, oa)
    gts = O.generate_genotypes(A, 2)
    sc = aa.run_filter(); so = O.run_filter(Mo, gts)
    assert np.abs(sc - so).max() <= 1e-9 * max(np.abs(so).max(), 1.0)
    for ms in (1, 7, 5000):
        fd = float(rng.choice([p.filt_diff, 5.0, 0.0]))
        k1 = api.truncate_ixs(sc, np.arange(len(sc)), fd, ms, p.threads)
        assert np.array_equal(aa.prefilter_truncate(fd, ms, p.threads), k1)
    # solver stages on the GPU's own tables
    n_good = aa.n_good()
    if n_good >= 20:
        st, w, unm, uk = aa.status(); off, pa = aa.pair_alns()
        ol.inject_tables(loc.depth_lut(), loc.window_weights())
        ob = O.alns_from_arrays(A, st, w, unm, off, pa)
        order = np.argsort(-sc, kind="stable")[: int(rng.integers(2, 12))]
        sub = gts[order]
        att = int(rng.integers(1, 4))
        seeds = api.chain_seeds(int(rng.integers(1, 1 << 20)), len(sub) * att)
        for kind in (cdefs.SOLVER_GREEDY, cdefs.SOLVER_ANNEAL):
            sv = api.default_solver(kind)
            if kind == cdefs.SOLVER_ANNEAL: sv.anneal_steps, sv.plato_size = int(rng.integers(200, 3000)), int(rng.integers(100, 2000))
            else: sv.sample_size, sv.plato_size, sv.best_start = int(rng.integers(1, 17)), int(rng.integers(10, 300)), int(rng.integers(0, 2))
            gm, gv, gl = api.solve_stage(aa, sub, sv, att, seeds)
            om, ov, olk = O.solve_stage(ol, ob, sub, sv, att, seeds)
            assert np.abs(gl - olk).max() <= 1e-9 * np.abs(olk).max(), (kind, np.abs(gl - olk).max())
    aa.close()
    return f"A={A} k={k} tech={tech} rl={rl} n={n} good={n_good}"


def recovery_case(ctx, rng, i):
    """alignment recovery on random haplotypes and reads with errors, clips, indels, either strand, single / paired ends"""
    from tests.test_oracle_transfer import make_haps
    from tests.test_gpu_parity import _recovery_case
    from tests.helpers import make_bg
    nh = int(rng.integers(2, 9))
    haps = make_haps(rng, nh, int(rng.integers(1500, 3200)))
    acgt = list(b"ACGT")
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    pairs = []
    for q in range(int(rng.integers(40, 200))):
        src = int(rng.integers(0, nh)); Lh = len(haps[src])
        p1 = int(rng.integers(0, Lh - 700)); p2 = min(Lh - 150, p1 + int(rng.integers(160, 480)))
        r1, r2 = bytearray(haps[src][p1:p1 + 150]), bytearray(haps[src][p2:p2 + 150])
        c1 = c2 = "150="
        kind = int(rng.integers(0, 6))
        if kind == 1:
            k = int(rng.integers(3, 30)); r1[:k] = bytes(rng.choice(acgt, k).tolist()); c1 = f"{k}S{150 - k}="
        elif kind == 2:
            x = int(rng.integers(20, 120)); k = int(rng.integers(1, 6))
            r1 = r1[:x] + bytearray(rng.choice(acgt, k).tolist()) + r1[x:150 - k]; c1 = f"{x}={k}I{150 - x - k}="
        elif kind == 3 and p1 + 160 < Lh:
            x = int(rng.integers(20, 120)); k = int(rng.integers(1, 6))
            r1 = bytearray(haps[src][p1:p1 + x] + haps[src][p1 + x + k:p1 + 150 + k]); c1 = f"{x}={k}D{150 - x}="
        elif kind == 4:
            x = int(rng.integers(5, 145)); r1[x] = ord("A") if r1[x] != ord("A") else ord("C"); c1 = f"{x}=1X{149 - x}="
        if rng.random() < 0.25:
            k = int(rng.integers(2, 20)); r2[150 - k:] = bytes(rng.choice(acgt, k).tolist()); c2 = f"{150 - k}={k}S"
        if rng.random() < 0.5:
            recs = [(src, p1, 0, c1), (src, p2, cdefs.FLAG_MATE2 | cdefs.FLAG_REVERSE, c2)]
        else:
            recs = [(src, p1, cdefs.FLAG_REVERSE, c1), (src, p2, cdefs.FLAG_MATE2, c2)]
        pairs.append({"seq1": bytes(r1).decode(), "seq2": bytes(r2).decode(), "recs": recs})
    tf = int(rng.choice([0, 1, 3, 100]))
    n_rec, aa, oa = _recovery_case(ctx, haps, pairs, make_bg(), tf=tf)
    aa.close()
    return f"haps={nh} pairs={len(pairs)} transfer_fails={tf} recovered={n_rec}"


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    ctx = api.Context(0)
    t0 = time.time()
    for i in range(n):
        try:
            msg = one_case(ctx, rng, i)
        except Exception:
            print(f"case {i}: FAILED", flush=True); traceback.print_exc(); sys.exit(1)
        print(f"case {i}: ok  {msg}  [{time.time() - t0:.0f} s]", flush=True)
    for i in range(n):
        try:
            msg = recovery_case(ctx, rng, i)
        except Exception:
            print(f"recovery case {i}: FAILED", flush=True); traceback.print_exc(); sys.exit(1)
        print(f"recovery case {i}: ok  {msg}  [{time.time() - t0:.0f} s]", flush=True)
    for seed in range(10, 10 + n):
        try:
            mapper_case(ctx, seed)
        except Exception:
            print(f"mapper seed {seed}: FAILED", flush=True); traceback.print_exc(); sys.exit(1)
    print(f"mapper: {n} random inputs ok  [{time.time() - t0:.0f} s]", flush=True)
    base = int(sys.argv[2]) * 1000 if len(sys.argv) > 2 else 100
    for seed in range(base, base + n):
        try:
            long_mapper_case(ctx, seed)
        except Exception:
            print(f"long-route mapper seed {seed}: FAILED", flush=True); traceback.print_exc(); sys.exit(1)
    print(f"long-route mapper: {n} random inputs ok  [{time.time() - t0:.0f} s]", flush=True)
    print("all ok")

main()
