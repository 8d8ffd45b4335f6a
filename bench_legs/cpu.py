"""The CPU baseline of the bench line: the oracle — the C restatement of the reference's algorithms — with the reference's own
thread structure, timed on the host cores of the GPU box. This is the ONLY module of the harness that touches tests.oracle_ffi
(the oracle is test infrastructure: a checker and a baseline, never part of what is measured or shipped)."""
import os
import subprocess
import time

import numpy as np

from locityper_amd import api
from .common import cpu_model, physical_cores, progress


def native_oracle(root):
    """the restatement compiled -O3 -march=native — so it is compiled HERE, on the box it is timed on (the portable
    build that travels with the repository is x86-64-v3), and tests.oracle_ffi is pointed at that build before its first use.
    Returns what the line says about the build."""
    from tests import oracle_ffi as O
    if O._lib is not None:
        return "the build that was loaded first: " + O.LIB_PATH
    try:
        subprocess.run(["make", "-B", "-C", os.path.join(root, "oracle"), "NATIVE=1"], check=True, capture_output=True, timeout=300)
        path = os.path.join(root, "oracle", "_build", "native", "liblcty_oracle.so")
        if os.path.exists(path):
            O.LIB_PATH = path
            return "gcc -O3 -march=native, built on this host at the start of the leg"
    except (OSError, subprocess.SubprocessError) as e:
        progress(f"  native build of the oracle failed ({e}); the portable build is timed")
    return "gcc -O3 -march=x86-64-v3 (the portable build: the native build failed on this host)"


def _median(xs):
    return float(np.median(xs))


def cpu_baseline(args, build, L, params, first, aa, gts, all_ixs, greedy, anneal, G, loc):
    """The reference's CPU path beside the GPU number, at threads = 8 (the reference default, genotype.rs:127) and at
    min(physical cores, 32):
      load     AllAlignments::load: single-threaded BAM loop + recover_and_group_alignments on `threads` workers, reads dealt
               round-robin (locs.rs:1116-1174) -> orc_load_mt on --cpu-sample read pairs of the same workload (default 262 144), taken
               as four equal slices spread over the batch; linear in the read pairs: scaled to the workload;
      filter   run_filter, single-threaded as solve.rs:87-122, MEASURED AT FULL SIZE: all G genotypes over the matrix of all good read
               pairs of the workload (the matrix the GPU scored: bit-identical to the oracle's by the parity tests) — unless the
               sample says that takes more than a minute, then scaled from the sample;
      solver   the stages of MainWorker::run (solve.rs:1047-1062: genotypes in contiguous runs over the workers) -> orc_solve_stage_mt
               on ALL read pairs of the workload (the oracle gets the batch the GPU scored). At the widest setting 64 greedy chains,
               median of --cpu-reps runs, and 64 annealing chains once; at 8 threads 8 + 8 chains once (a full-size chain is seconds).
    The whole path on the workload = load (scaled) + run_filter (measured) + 5 000 greedy + 20 x 20 annealing chains (the default
    scheme) at the measured chain rates: the chain counts timed are in the entry, the scaling is stated, and no ratio to the GPU
    figure is put into the line. `build`: what native_oracle() said about the library that is timed."""
    from tests import oracle_ffi as O
    A = args.alleles
    ns = min(args.cpu_sample, args.pairs) // 4 * 4
    per = max(ns // 4, 1)
    slices = []
    for q in range(4):
        lo = (args.pairs * q // 4) // 32 * 32
        if q == 0 and per <= first.n_pairs:
            slices.append(first.slice(0, per))
        elif lo + per <= args.pairs:
            slices.append(L.reads(lo, per))
    ns = sum(c.n_pairs for c in slices)
    ol = O.OracleLocus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, params)
    reps = max(1, args.cpu_reps)
    n_phys = physical_cores()
    # the second setting stops at 32 threads: beyond that the oracle's stage loop does not scale (run r4_v1, 128 cores: 2.72 chains/s
    # against 2.01 at 8 threads — every worker streams the locus' 6 GB of pair-alignments per genotype, GenotypeAlignments::new)
    n_max = min(n_phys, 32)
    thread_sets = [8] if n_max <= 8 else [8, n_max]
    # ---- run_filter: single thread whatever `threads` is; first on a sample (for the estimate), then at full size ----
    probe = slices[0].slice(0, min(slices[0].n_pairs, 65536))
    oa = ol.load(probe)
    Mo = oa.best_aln_matrix()
    tc = time.perf_counter()
    so = O.run_filter(Mo, gts)
    O.truncate(so, all_ixs, params.filt_diff, 5000, 8)
    t_filter_sample = time.perf_counter() - tc
    n_good_sample, filter_pairs = oa.n_good, probe.n_pairs
    t_filter_full = t_filter_sample * args.pairs / filter_pairs
    filter_measured = False
    if t_filter_full <= 60.0:
        Mfull = aa.best_aln_matrix()                                        # [A][n_good] as locs.rs:1203-1212 lays it out
        tc = time.perf_counter()
        so = O.run_filter(Mfull, gts)
        O.truncate(so, all_ixs, params.filt_diff, 5000, 8)
        t_filter_full = time.perf_counter() - tc
        filter_measured = True
        del Mfull
    del oa, Mo
    # ---- solver inputs at full size: the scored batch of the GPU ----
    solver_pairs, oa_full = 0, None
    if not args.no_solve:
        try:
            avail_kb = int(next(ln for ln in open("/proc/meminfo") if ln.startswith("MemAvailable")).split()[1])
        except (OSError, StopIteration):
            avail_kb = 0
        status, weight, unm, _ = aa.status()
        off, pa = aa.pair_alns()
        need_kb = 3 * pa.nbytes // 1024
        if avail_kb and avail_kb < need_kb:
            raise RuntimeError(f"cpu_baseline: {need_kb >> 20} GB of host memory needed for the solver sample, {avail_kb >> 20} GB "
                               "available (use --cpu-sample 0 to skip)")
        oa_full = O.alns_from_arrays(A, status, weight, unm, off, pa)
        solver_pairs = aa.n_pairs
        del pa, off
    order = np.argsort(-so, kind="stable")
    by = {}
    for T in thread_sets:
        progress(f"  CPU: load on {ns} read pairs at {T} threads")
        t_all = s_all = g_all = 0.0
        for c in slices:
            tc = time.perf_counter()
            ob, secs = ol.load_mt(c, T)
            ob.best_aln_matrix()
            t_all += time.perf_counter() - tc
            s_all += secs[0]
            g_all += secs[1]
            del ob
        scale = args.pairs / ns
        entry = {"threads": T, "load_s": t_all, "load_serial_s": s_all, "load_group_s": g_all, "load_read_pairs": ns,
                 "run_filter_s": t_filter_full, "run_filter_measured_at_full_size": filter_measured,
                 "reads_scored_per_s": args.pairs / (t_all * scale + t_filter_full)}
        total = t_all * scale + t_filter_full
        if oa_full is not None:
            widest = T == thread_sets[-1]
            ng = min(64 if widest else 8, len(order))
            na = min(64 if widest else 8, len(order))
            progress(f"  CPU: {ng} greedy chains x {reps if widest else 1} + {na} annealing chains on {solver_pairs} read pairs at {T} threads")
            tgreedy = []
            for rep in range(reps if widest else 1):
                tc = time.perf_counter()
                O.solve_stage(ol, oa_full, gts[order[:ng]], greedy, 1, api.chain_seeds(1000 + rep, ng), threads=T)
                tgreedy.append(time.perf_counter() - tc)
            tc = time.perf_counter()
            O.solve_stage(ol, oa_full, gts[order[:na]], anneal, 1, api.chain_seeds(2000, na), threads=T)
            t_anneal = time.perf_counter() - tc
            g_cps, a_cps = ng / _median(tgreedy), na / t_anneal
            entry.update({"greedy_chains_per_s": g_cps, "anneal_chains_per_s": a_cps, "solver_read_pairs": solver_pairs,
                          "greedy_chains_timed": ng, "greedy_repetitions": len(tgreedy), "anneal_chains_timed": na,
                          "greedy_chains_per_s_per_thread": g_cps / min(T, ng), "anneal_chains_per_s_per_thread": a_cps / min(T, na),
                          "chains_per_s": 5400.0 / (5000.0 / g_cps + 400.0 / a_cps)})
            total += (5000.0 / g_cps + 400.0 / a_cps) * (args.pairs / solver_pairs)
        entry["seconds_per_locus"] = total
        entry["value"] = args.pairs / total
        entry["note"] = f"{T} threads of {n_phys} physical cores"
        by[f"threads_{T}"] = entry
    # the reported baseline is the FASTER of the thread settings tried, named by its thread count
    best_key = max(by, key=lambda k_: by[k_]["value"])
    # ---- the oracle's chains against the GPU's, on the full batch (stoch.rs:81-120, 195-245): the timed runs above evaluate BayesCalc
    # on the fly beyond depth 256 as the reference does (own lgamma: a near-tie can flip); for the comparison the oracle gets the
    # device's tables, so a chain has to follow the same moves and the likelihoods agree to 1e-9 relative
    chains_check = None
    if oa_full is not None:
        ol.inject_tables(loc.depth_lut(), loc.window_weights())
        ol.inject_depth_table(loc.depth_table(8192))
        nchk = min(8, len(order))
        sub = gts[order[:nchk]]
        worst = 0.0
        for solver, master in ((greedy, 3000), (anneal, 4000)):
            seeds = api.chain_seeds(master, nchk)
            _, _, gl = api.solve_stage(aa, sub, solver, 1, seeds)
            _, _, olk = O.solve_stage(ol, oa_full, sub, solver, 1, seeds, threads=min(8, n_phys))
            worst = max(worst, float(np.abs(gl - olk).max() / np.abs(olk).max()))
        chains_check = {"greedy_chains": nchk, "anneal_chains": nchk, "read_pairs": solver_pairs, "max_relative_difference": worst,
                        "chains_equal_oracle": bool(worst <= 1e-9)}
    best = by[best_key]
    sample = (f"load on {ns} read pairs x {A} alleles (four slices spread over the batch), scaled to {args.pairs}; run_filter "
              + (f"measured on all {args.pairs} read pairs, all {G} genotypes, one thread as upstream; " if filter_measured
                 else f"on {filter_pairs} read pairs ({n_good_sample} good), scaled; ")
              + (f"solver chains on all {solver_pairs} read pairs (inputs = the batch the GPU scored): "
                 f"{best.get('greedy_chains_timed')} greedy (median of {best.get('greedy_repetitions')}) and "
                 f"{best.get('anneal_chains_timed')} annealing chains timed; " if oa_full is not None else "")
              + "whole path = load + run_filter + 5 000 greedy + 400 annealing chains at the measured rates; the faster of the thread "
                "settings in by_threads")
    return {"value": best["value"], "unit": "read pairs/s", "cores": best["threads"], "kind": "port", "sample": sample,
            "oracle_build": build, "cpu_model": cpu_model(), "physical_cores": n_phys, "cpu_count": os.cpu_count(),
            "by_threads": by, "reported_setting": best_key, "chains_check": chains_check,
            "reads_scored_per_s": best["reads_scored_per_s"], "chains_per_s": best.get("chains_per_s"),
            "note": "reference-algorithm CPU restatement (oracle/), never 'locityper': the Rust reference cannot be built here. A baseline, "
                    "not a target: no ratio to it is reported"}


def recruitment_baseline(L, rprm, n_pairs=20000):
    """recruit_read_pair of the oracle on one core: random 150 + 150-base pairs against the locus' alleles."""
    from tests import oracle_ffi as O
    ot = O.OracleTargets(rprm.minimizer_k, rprm.minimizer_w, rprm.match_frac, rprm.match_length, rprm.thresh_kmer_count)
    ot.add_locus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k)
    ot.finalize()
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    words = np.random.default_rng(11).integers(0, 1 << 32, size=n_pairs * 20, dtype=np.uint64).astype(np.uint32).reshape(n_pairs, 2, 10)
    codes = ((words[..., None] >> (2 * np.arange(16, dtype=np.uint32))) & 3).reshape(n_pairs, 2, 160)[:, :, :150]
    sq = acgt[codes]
    tc = time.perf_counter()
    for i in range(n_pairs):
        ot.recruit(sq[i, 0].tobytes(), sq[i, 1].tobytes())
    return n_pairs / (time.perf_counter() - tc)


def long_read_recovery_baseline(Ls, ps, n_alleles, n_reads=8):
    """The CPU beside the long-read legs: the oracle has no restatement of the mapper in C (tests/pyref_map_long.py is Python); what it
    has for long reads is the reference's own route once a mapper has placed a read — AllAlignments::load with alignment recovery onto
    the other alleles (locs.rs:1085-1185, transfer.rs:70-140) — timed on one core."""
    from tests import oracle_ffi as O
    ols = O.OracleLocus(Ls.seqs, Ls.seq_off, Ls.counts, Ls.cnt_off, Ls.k, Ls.bg, ps)
    hos = O.HapAlns(n_alleles, transfer_fails=100, max_div=0.1)
    for q, r, w, _, _ in Ls.hap_alns():
        hos.add(q, r, w)
    hos.sort()
    prim = Ls.reads(0, n_reads, primaries_only=True)
    tc = time.perf_counter()
    oas = ols.load_recover(prim, hos)
    dtc = time.perf_counter() - tc
    return {"value": n_reads / dtc, "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": f"{n_reads} of those reads with the generator's primary record: oracle AllAlignments::load with alignment recovery onto "
                      f"the other {n_alleles - 1} alleles (the reference's route behind its mapper; the mapper itself has no C restatement)",
            "alignments_per_s": n_reads * n_alleles / dtc, "good_reads": oas.n_good}


def whole_path_chains_check(L, p, loc, aa, gts, scores, greedy, anneal, n_chains=8):
    """Eight greedy and eight annealing chains of the oracle against the device's on the batch the device scored (any technology):
    the oracle gets the device's products and tables, so a chain has to make the same moves."""
    from tests import oracle_ffi as O
    A = loc.n_alleles
    ol = O.OracleLocus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    ol.inject_tables(loc.depth_lut(), loc.window_weights())
    ol.inject_depth_table(loc.depth_table(1 << 15))
    status, weight, unm, _ = aa.status()
    off, pa = aa.pair_alns()
    oa = O.alns_from_arrays(A, status, weight, unm, off, pa)
    order = np.argsort(-scores, kind="stable")[:n_chains]
    sub = gts[order]
    worst = 0.0
    for solver, master in ((greedy, 3000), (anneal, 4000)):
        seeds = api.chain_seeds(master, len(sub))
        _, _, gl = api.solve_stage(aa, sub, solver, 1, seeds)
        _, _, olk = O.solve_stage(ol, oa, sub, solver, 1, seeds, threads=min(8, physical_cores()))
        worst = max(worst, float(np.abs(gl - olk).max() / np.abs(olk).max()))
    return {"greedy_chains": len(sub), "anneal_chains": len(sub), "reads": int(aa.n_pairs), "max_relative_difference": worst,
            "chains_equal_oracle": bool(worst <= 1e-9)}


def exact_against_highs(state, n_genotypes):
    """Beside the exact-solver leg (bench_legs/short_reads.py::exact_solver_leg), on the host: HiGHS itself (scipy.optimize.milp) on the
    reference's integer programme (HighsSolver::define_model, highs.rs:38-100) built from the ORACLE's GenotypeAlignments with the same
    tweak (tests/pyref_highs.py) — what the reference's back end takes for the same model, one thread as the reference runs a model, and
    how far below its optimum the library's answer is. For the best `n_genotypes` of the prefilter."""
    import ctypes as C
    import time
    from tests import oracle_ffi as O, pyref_highs as H
    L, p, loc, aa, every, seeds, lik = state
    ol = O.OracleLocus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    ol.inject_tables(loc.depth_lut(), loc.window_weights())
    st, w8, unm, _ = aa.status()
    off, pa = aa.pair_alns()
    oa = O.alns_from_arrays(8, st, w8, unm, off, pa)
    lib = O.lib()
    lib.orc_depth_ln_prob.restype = C.c_double
    rows = []
    for r in range(min(n_genotypes, len(every))):
        g = O.OracleGtAlns(ol, oa, tuple(int(x) for x in every[r]))
        g.apply_tweak(int(seeds[r]))
        a = g.arrays()
        gc, ww = g.window_distr()
        t1 = time.perf_counter()
        ok, h_assgn, _, info = H.solve(a["read_ixs"], a["ln_prob"], a["windows"], gc, ww,
                                       lambda x, d: lib.orc_depth_ln_prob(ol._h, int(gc[x]), float(ww[x]), int(d)),
                                       1.0 - p.lik_skew, 1.0 + p.lik_skew, time_limit=120.0)
        h_lik = g.likelihood(h_assgn)[0] if ok else None
        rows.append({"genotype": [int(x) for x in every[r]], "highs_seconds": time.perf_counter() - t1, "highs_status_optimal": bool(ok),
                     "highs_nodes": info.get("mip_node_count"), "highs_likelihood": h_lik, "library_likelihood": float(lik[r]),
                     "library_below_highs_relative": (None if h_lik is None else float((h_lik - lik[r]) / abs(h_lik)))})
    return {"what": "scipy.optimize.milp (HiGHS) on HighsSolver::define_model's programme from the oracle's GenotypeAlignments, the reference's "
                    "options, one host thread", "rows": rows}
