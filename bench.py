#!/usr/bin/env python3
"""bench.py — reads/s scored + genotypes/s solved on the BASELINE.json config
("1M synthetic 150 bp PE reads, 1 locus, 256 alleles, k=25" = configs[1]).

A step = one pass of the hot path over one locus' batch, inputs resident in HBM:
    lcty_score_reads   (K2+K4+K5+K7+K8: AllAlignments::load -> likelihood matrix + pair alignments)
    lcty_prefilter_async (K9: run_filter over all C(A+1,2) genotypes), scores stay in HBM
    lcty_prefilter_truncate (K10: truncate_ixs as a device sort + prefix; the kept indices D2H)
    lcty_solve_stage   (K11-K14: greedy on the survivors, 1 attempt; annealing on the best 20, 20 attempts)
    lcty_discard_improbable / lcty_produce_result (K15: final genotype comparison)
N > 1: one process per GPU, one independent locus per rank (loci are independent in the reference,
command/genotype.rs:1331-1351) -> weak scaling, no data-path collective; torch.distributed (gloo)
only carries the barrier and the max-over-ranks of the timed region.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from locityper_amd import _lib, api, synth, cdefs  # noqa: E402

# SURVEY.md §8(d) / BASELINE.md algorithmic bytes per read pair scored (config 2, f64 matrix):
# 75 B packed bases + 2*A*16 B alignment table + 252*8 B k-mer probe slots + A*8 B matrix row
ALG_BYTES_FIXED = 75 + 2016


from scripts.sources_sha import sources_sha16


def survey_bytes_per_pair(n_alleles):
    return ALG_BYTES_FIXED + 2 * n_alleles * 16 + n_alleles * 8


HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16, help="loci of the timed queue (its last locus finishes alone: 0.33 s of drain shared by all)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=1_000_000, help="read pairs per locus (BASELINE: 1M)")
    ap.add_argument("--alleles", type=int, default=256)
    ap.add_argument("--chunk", type=int, default=32768, help="pairs per generated/uploaded chunk")
    ap.add_argument("--knob", action="append", default=[], help="developer experiments: name=value for lcty_ctx_set_knob (repeatable)")
    ap.add_argument("--diag", action="store_true", help="load the developer build of the library (make -C locityper_amd/csrc DIAG=1): trace and timing knobs exist there only")
    ap.add_argument("--cpu-sample", type=int, default=65536, help="pairs given to the CPU baseline (0 = skip)")
    ap.add_argument("--ont-sample", type=int, default=6144,
                    help="reads of the extra long-read measurement (BASELINE.json configs[2] shape: 10 kb ONT reads x the locus' alleles, "
                         "the mapper reports the primaries, the other alleles are reached by alignment recovery); 0 = skip")
    ap.add_argument("--many-alleles-sample", type=int, default=65536,
                    help="read pairs of the extra measurement at 4 096 alleles (BASELINE.json configs[4], one GPU's shard): scoring and the "
                         "prefilter as f64 tile kernel and as integer Gram contraction on the matrix cores; 0 = skip")
    ap.add_argument("--format", choices=("counted", "records"), default="counted",
                    help="how the alignment table reaches the library: 16-byte counted alignments (lcty_reads_append_counted, SURVEY 8(d)'s "
                         "alignment-table entry; the default) or BAM records with their CIGAR words (lcty_reads_append)")
    ap.add_argument("--cpu-reps", type=int, default=3, help="repetitions of every CPU-baseline figure (the median is reported)")
    ap.add_argument("--traffic", default=os.path.join(ROOT, "profiles", "r05_pmc_traffic.json"),
                    help="per-launch HBM bytes from the PMC passes (scripts/pmc_summary.py); used when it matches the workload")
    ap.add_argument("--no-solve", action="store_true", help="leave the solver stages out of the step (score + prefilter only)")
    ap.add_argument("--shard-reads", action="store_true",
                    help="one locus over all ranks, whole path (BASELINE configs[4]): every rank scores and prefilters a contiguous shard of the "
                         "read pairs, the run_filter scores are SUM-all-reduced on the devices (RCCL), truncate_ixs runs everywhere, every solver "
                         "stage all-gathers the location-table rows of its alleles and deals its chains to the ranks "
                         "(lcty_solve_stage_read_sharded); strong scaling. Not the default: the driver's runs are one locus per rank")
    ap.add_argument("--shard-chains", action="store_true",
                    help="one locus over all ranks, whole path: every rank scores and prefilters all reads of the locus (replicated), the "
                         "(genotype, attempt) chains of both solver stages are dealt to the ranks and their likelihoods all-gathered on the "
                         "devices (RCCL; lcty_solve_stage_sharded, SURVEY 8e level 3); strong scaling. Not the default either")
    ap.add_argument("--distinct-loci", type=int, default=2,
                    help="extra measurement: a queue of loci that are NOT resident — this many distinct loci in page-locked host memory, every "
                         "position of the queue uploaded (lcty_reads_append_counted on the copy stream, from a loader thread) while the position "
                         "before it is solved, three batch objects rotating (lcty_solve_queue_fed); 0 = skip. The default, 2, takes the two loci "
                         "of the main measurement, so that the two queues do the same work (loci differ: with a third locus the resident queue "
                         "itself goes from 539 to 621 ms per step)")
    ap.add_argument("--distinct-steps", type=int, default=0, help="positions of the timed queue of the --distinct-loci measurement (0: as many as --steps, so that the two queues compare like for like)")
    ap.add_argument("--loci-seeds", default="", help="developer measurement: seed offsets of the resident loci, comma-separated (default 0,1)")
    ap.add_argument("--distinct-no-upload", action="store_true", help="developer measurement: the rotation of three batch objects through lcty_solve_queue_fed "
                    "WITHOUT the uploads (every batch keeps the locus the warm-up gave it): what the rotation alone costs")
    ap.add_argument("--oversubscribe", action="store_true", help="allow more ranks than devices (launch-path checks on a one-GPU box; reported in the line)")
    ap.add_argument("--pipeline", type=int, default=0, help="ignored (round 1 option; the queue of loci is the default mode now)")
    ap.add_argument("--recruit-sample", type=int, default=8_000_000,
                    help="read pairs of the extra recruitment measurement (the step before the path, SURVEY 8f rank 1; 0 = skip)")
    ap.add_argument("--ont-map-sample", type=int, default=2048, help="of the --ont-sample reads: mapped from their bases alone onto every allele (long route of candidate generation), then scored and prefiltered (0 = skip)")
    ap.add_argument("--map-sample", type=int, default=32768, help="read pairs mapped onto 8 basis alleles by the candidate-generation slice (0 = skip)")
    ap.add_argument("--ont-stream-sample", type=int, default=65536,
                    help="10-kb ONT reads of the configs[2] leg from bases alone, streamed (mapped onto all alleles on the device, scored, prefiltered; 0 = skip)")
    ap.add_argument("--recovery-sample", type=int, default=262144,
                    help="read pairs of the extra alignment-recovery measurement (K6, outside the timed region; 0 = skip)")
    return ap.parse_args()


def physical_cores():
    """Physical cores of the host (unique (physical id, core id) pairs of /proc/cpuinfo); falls back to os.cpu_count()."""
    try:
        cores, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
        n = len(cores) or os.cpu_count()
    except OSError:
        n = os.cpu_count()
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    return max(1, int(n))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


_T_START = time.time()


def progress(what):
    """where the wall time of a run goes (stderr; the JSON line is the only thing on stdout)"""
    print(f"[bench {time.time() - _T_START:7.1f} s] {what}", file=sys.stderr, flush=True)


def cpu_baseline(args, L, params, first, aa, gts, all_ixs, greedy, anneal, G, loc):
    """The reference's CPU path beside the GPU number: the oracle — a C restatement of the reference algorithms —
    with the reference's own thread structure, at threads = 8 (the reference default, genotype.rs:127) and at all physical cores:
      load     AllAlignments::load: single-threaded BAM loop + recover_and_group_alignments on `threads` workers, reads dealt
               round-robin (locs.rs:1116-1174) -> orc_load_mt on --cpu-sample read pairs of the same workload, taken as four equal
               slices spread over the batch (pairs 0.., P/4.., P/2.., 3P/4..); linear in the read pairs: scaled to the workload;
      filter   run_filter, single-threaded as solve.rs:87-122, MEASURED AT FULL SIZE: all G genotypes over the matrix of all good read
               pairs of the workload (the matrix the GPU scored: bit-identical to the oracle's by the parity tests) — unless the
               sample says that takes more than a minute, then scaled from the sample;
      solver   the stages of MainWorker::run (solve.rs:1047-1062: genotypes in contiguous runs over the workers) -> orc_solve_stage_mt
               on ALL read pairs of the workload (the oracle gets the batch the GPU scored), one genotype per worker and stage:
               `threads` greedy chains, then `threads` annealing chains, every worker refilling ONE GenotypeAlignments object
               (oracle/lcty_oracle_solve.c: orc_gt_alns_fill); chains per second and per thread are both in the entry.
    The whole path on the workload = load (scaled) + run_filter (measured) + 5 000 greedy + 20 x 20 annealing chains (the default
    scheme, from the measured chain rates); median of --cpu-reps runs where a figure is repeated."""
    from tests import oracle_ffi as O
    A = args.alleles
    ns = min(args.cpu_sample, first.n_pairs * 4, args.pairs) // 4 * 4
    per = max(ns // 4, 1)
    slices = [first.slice(0, min(per, first.n_pairs))]
    for q in (1, 2, 3):
        lo = (args.pairs * q // 4) // 32 * 32
        if ns >= 4 and lo + per <= args.pairs: slices.append(L.reads(lo, per))
    ns = sum(c.n_pairs for c in slices)
    ol = O.OracleLocus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, params)
    reps = max(1, args.cpu_reps)
    med = lambda xs: float(np.median(xs))
    n_phys = physical_cores()
    # the second setting stops at 32 threads: beyond that the oracle's stage loop does not scale (run r4_v1, 128 cores: 2.72 chains/s against
    # 2.01 at 8 threads — every worker streams the locus' 6 GB of pair-alignments per genotype, GenotypeAlignments::new — and 128 full-size
    # chains at once are 95 s of the run)
    n_max = min(n_phys, 32)
    thread_sets = [8] if n_max <= 8 else [8, n_max]
    # ---- run_filter: single thread whatever `threads` is; first on the first slice (for the estimate), then at full size ----
    oa = ol.load(slices[0])
    Mo = oa.best_aln_matrix()
    tc = time.perf_counter()
    so = O.run_filter(Mo, gts)
    O.truncate(so, all_ixs, params.filt_diff, 5000, 8)
    t_filter_sample = time.perf_counter() - tc
    n_good_sample = oa.n_good
    filter_pairs = slices[0].n_pairs
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
    # ---- solver inputs at full size: the scored batch of the GPU ----
    solver_pairs, oa_full = 0, None
    if not args.no_solve:
        try:
            avail_kb = int(next(l for l in open("/proc/meminfo") if l.startswith("MemAvailable")).split()[1])
        except (OSError, StopIteration):
            avail_kb = 0
        status, weight, unm, _ = aa.status()
        off, pa = aa.pair_alns()
        need_kb = 3 * pa.nbytes // 1024
        if avail_kb and avail_kb < need_kb:
            raise RuntimeError(f"cpu_baseline: {need_kb >> 20} GB of host memory needed for the solver sample, {avail_kb >> 20} GB available "
                               "(use --cpu-sample 0 to skip)")
        oa_full = O.alns_from_arrays(A, status, weight, unm, off, pa)
        solver_pairs = aa.n_pairs
        del pa, off
    by = {}
    for T in thread_sets:
        tl, ts_, tg_ = [], [], []
        for _ in range(reps):
            t_all = s_all = g_all = 0.0
            for c in slices:
                tc = time.perf_counter()
                ob, secs = ol.load_mt(c, T)
                ob.best_aln_matrix()
                t_all += time.perf_counter() - tc; s_all += secs[0]; g_all += secs[1]
                del ob
            tl.append(t_all); ts_.append(s_all); tg_.append(g_all)
        t_load = med(tl)
        scale = args.pairs / ns
        entry = {"threads": T, "load_s": t_load, "load_serial_s": med(ts_), "load_group_s": med(tg_), "load_read_pairs": ns,
                 "run_filter_s": t_filter_full, "run_filter_measured_at_full_size": filter_measured,
                 "reads_scored_per_s": args.pairs / (t_load * scale + t_filter_full)}
        total = t_load * scale + t_filter_full
        if oa_full is not None:
            # the best genotypes of the prefilter: what the stages would work on
            order = np.argsort(-so, kind="stable")
            ng = min(max(T, 8), len(order))
            sub_g = gts[order[:ng]]
            na = min(max(T, 4), len(order))
            sub_a = gts[order[:na]]
            tgreedy, tanneal = [], []
            for rep in range(1):                                              # a full-size chain per worker is seconds: once
                tc = time.perf_counter()
                O.solve_stage(ol, oa_full, sub_g, greedy, 1, api.chain_seeds(1000 + rep, ng), threads=T)
                tgreedy.append(time.perf_counter() - tc)
                tc = time.perf_counter()
                O.solve_stage(ol, oa_full, sub_a, anneal, 1, api.chain_seeds(2000 + rep, na), threads=T)
                tanneal.append(time.perf_counter() - tc)
            g_cps, a_cps = ng / med(tgreedy), na / med(tanneal)
            entry.update({"greedy_chains_per_s@R": g_cps, "anneal_chains_per_s@R": a_cps, "solver_read_pairs": solver_pairs,
                          "greedy_chains_timed": ng, "anneal_chains_timed": na,
                          "greedy_chains_per_s_per_thread": g_cps / min(T, ng), "anneal_chains_per_s_per_thread": a_cps / min(T, na),
                          "chains_per_s": 5400.0 / (5000.0 / g_cps + 400.0 / a_cps)})
            total += (5000.0 / g_cps + 400.0 / a_cps) * (args.pairs / solver_pairs)
        entry["seconds_per_locus"] = total
        entry["value"] = args.pairs / total
        by[f"threads_{T}"] = entry
    # the reported baseline is the FASTER of the thread settings tried (8 = the reference's default; min(physical cores, 32): beyond that the
    # restatement's stage loop stops scaling, see above) — named by its thread count, not "all cores"
    best_key = max(by, key=lambda k_: by[k_]["value"])
    for k_ in by: by[k_]["note"] = f"{by[k_]['threads']} threads of {n_phys} physical cores"
    widest = by[f"threads_{thread_sets[-1]}"]
    if oa_full is not None and len(thread_sets) > 1 and "chains_per_s" in widest:
        # how the stage loop scales from the reference's default of 8 threads to the widest setting (1.0 = linear in the threads)
        t8 = by["threads_8"]
        widest["chain_scaling_vs_8_threads"] = (widest["chains_per_s"] / t8["chains_per_s"]) / max(widest["threads"] / 8.0, 1.0)
    # ---- the oracle's chains against the GPU's, on the full batch (stoch.rs:81-120, 195-245): the timed runs above evaluate BayesCalc on
    # the fly beyond depth 256 as the reference does (own lgamma: a near-tie can flip); for the comparison the oracle gets the device's
    # tables, so a chain has to follow the same moves and the likelihoods agree to 1e-9 relative
    chains_check = None
    if oa_full is not None:
        ol.inject_tables(loc.depth_lut(), loc.window_weights())
        ol.inject_depth_table(loc.depth_table(8192))
        order = np.argsort(-so, kind="stable")
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
    return {"value": best["value"], "unit": "read pairs/s", "cores": best["threads"], "kind": "port",
            "sample": f"load on {ns} read pairs x {A} alleles (four slices spread over the batch), scaled to {args.pairs}; run_filter "
                      + (f"measured on all {args.pairs} read pairs, all {G} genotypes, one thread as upstream; " if filter_measured else f"on {filter_pairs} read pairs ({n_good_sample} good), scaled; ")
                      + (f"solver chains on all {solver_pairs} read pairs (inputs = the batch the GPU scored); " if oa_full is not None else "")
                      + f"whole path = load + run_filter + 5 000 greedy + 400 annealing chains at the measured rates; median of {reps}; the faster of the thread settings in by_threads",
            "cpu_model": cpu_model(), "physical_cores": n_phys, "cpu_count": os.cpu_count(),
            "by_threads": by, "reported_setting": best_key, "chains_check": chains_check,
            "reads_scored_per_s": best["reads_scored_per_s"], "chains_per_s": best.get("chains_per_s"),
            "note": "reference-algorithm CPU restatement (oracle/), never 'locityper': the Rust reference cannot be built here"}


def spawn_ranks(args):
    """`python3 bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): this process becomes the launcher — it starts N
    fresh children of this very command line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, BEFORE it has loaded the
    HIP library or made any GPU call (a process that has touched the GPU never execs another program), relays rank 0's JSON line and
    exits non-zero when any child fails. The children are what `torchrun --nproc-per-node N` would have started."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                LCTY_BENCH_LAUNCH="self-spawned children of bench.py")
    # every rank's OpenMP teams (synthetic data, CSR validation), loader and validation threads get their share of the host's cores:
    # N ranks with the default "all cores" each would oversubscribe the host N-fold during set-up and inside the loader threads
    share = max(1, physical_cores() // max(args.gpus, 1))
    if "OMP_NUM_THREADS" not in os.environ: base["OMP_NUM_THREADS"] = str(share)
    base["LCTY_BENCH_HOST_THREADS"] = str(share)
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(args.gpus):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    # rank 0's line is read by a thread; the launcher polls ALL children: one that dies before the rendezvous would leave the others
    # waiting for it for ever — the rest is ended and the launcher exits non-zero as soon as any child fails
    import threading
    import time as _time
    got = {}
    reader = threading.Thread(target=lambda: got.setdefault("line", procs[0].stdout.read()), daemon=True)
    reader.start()
    codes = [None] * len(procs)
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None: codes[i] = p.poll()
        if any(c not in (None, 0) for c in codes):
            _time.sleep(2.0)                                       # the others may be on their way out with the same error
            for i, p in enumerate(procs):
                if codes[i] is None and p.poll() is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if codes[i] is None:
                    try: codes[i] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired: p.kill(); codes[i] = p.wait()
            break
        _time.sleep(0.2)
    if any(codes):
        print(f"bench.py: child ranks exited with {codes}", file=sys.stderr)
        sys.exit(next(c for c in codes if c) or 1)
    reader.join(timeout=30)
    sys.stdout.write(got.get("line", ""))
    sys.stdout.flush()
    sys.exit(0)


def long_route_traffic(n_alignments):
    """HBM bytes of the long route's align kernel for that many alignments of 10-kb reads, from the committed counter passes (None without the file)."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03_pmc_map_long_2048_ont_reads_x16.json")) as f:
            t = json.load(f)["traffic"]["align_kernel_bytes_per_alignment"]
        return (t["fetch_raw"] + t["write"]) * n_alignments
    except (OSError, KeyError, ValueError):
        return None


def distinct_loci_leg(args, ctx, loci, batches, stages, gts, resident_ms_per_step, host_chunks):
    """The queue as `locityper genotype` meets it: every locus arrives from the host. D distinct loci — the loci of the main measurement first,
    so that with the default D = 2 the two queues do the same work — (their counted alignment tables
    and bases in page-locked memory: 8.3 GB each at 1 M x 256), K positions cycling over them; a loader thread resets one of three batch
    objects (lcty_reads_reset), uploads the position's chunks (lcty_reads_append_counted: copies on the context's copy stream, the CSR
    validation on the host's cores next to them) and hands it to lcty_solve_queue_fed, which releases a batch when its last stage is
    done. Timed: K positions, the first upload included."""
    import threading
    D, K, A = args.distinct_loci, (args.distinct_steps or args.steps), args.alleles
    trace = any(kv.startswith("queue_trace=") and not kv.endswith("=0") for kv in args.knob)
    for b in batches:                                          # the resident loci of the main measurement make room
        b.close()
    ctx.trim()
    t0 = time.time()
    host = []                                                  # per locus: (SynthLocus, Locus, [(pinned chunk, pinned counted alignments)])
    n_chunks = (args.pairs + args.chunk - 1) // args.chunk
    from locityper_amd.cdefs import ReadsChunk, ALN_REC_DTYPE
    none_recs, none_cig = np.zeros(0, dtype=ALN_REC_DTYPE), np.zeros(0, dtype=np.uint32)
    caps = None
    up_bytes = 0
    for j in range(D):
        if j < len(loci): L, loc = loci[j]
        else:
            L = synth.SynthLocus(A, args.pairs, seed=synth.SEED + 100 + j)
            loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, api.resolve_params(api.default_params(), L.bg))
        chunks = []
        tb = tr = 0
        for ci in range(n_chunks):
            lo = ci * args.chunk
            ch = host_chunks[j][ci] if j < len(host_chunks) and host_chunks[j] else L.reads(lo, min(args.chunk, args.pairs - lo))
            alns = ctx.pinned_like(ch.counted(loc.allele_len))
            pc = ReadsChunk(*(ctx.pinned_like(a) for a in (ch.mate_len, ch.mate_off, ch.bases2, ch.nmask, ch.aln_off)), none_recs,
                            np.zeros(ch.n_pairs + 1, dtype=np.uint64), none_cig)
            chunks.append((pc, alns))
            tb += ch.n_bases; tr += len(ch.recs)
            if j == 0: up_bytes += alns.nbytes + pc.bases2.nbytes + pc.nmask.nbytes + pc.mate_len.nbytes + pc.mate_off.nbytes + pc.aln_off.nbytes
            if j < len(host_chunks) and host_chunks[j]: host_chunks[j][ci] = None
            del ch
        caps = (max(caps[0], tb), max(caps[1], tr)) if caps else (tb, tr)
        host.append((L, loc, chunks))
    ctx.set_knob("arena_cap_pct", 35)                         # one PairAlignment per (pair, allele) is the rule here; the bound is two per record
    cap_bases = (int(caps[0] * 1.01) + 1024) // 32 * 32 + 32
    rot = [api.AllAlignments(host[0][1], args.pairs, cap_bases, int(caps[1] * 1.01) + 4096, 0) for _ in range(3)]
    ctx.set_knob("arena_cap_pct", -1)
    setup_s = time.time() - t0

    def run(k, first_it):
        ready = [threading.Event() for _ in range(k)]
        free = [threading.Semaphore(1) for _ in range(3)]
        problems = []
        load_s = [0.0] * k

        def loader():
            try:
                for i in range(k):
                    free[i % 3].acquire()
                    tl = time.perf_counter()
                    L, loc, chunks = host[(first_it + i) % D]
                    b = rot[i % 3]
                    if not (args.distinct_no_upload and first_it == 0 and loaded_once[0]):
                        b.reset(loc)
                        for pc, alns in chunks:
                            b.append(pc, counted=alns)
                    load_s[i] = time.perf_counter() - tl
                    if trace: progress(f"  position {i}: loaded in {load_s[i]:.3f} s")
                    ready[i].set()
            except BaseException as e:                         # the queue must not wait for ever
                problems.append(e)
                for ev in ready: ev.set()

        def acquire(i):
            tw = time.perf_counter()
            ready[i].wait()
            if trace: progress(f"  position {i}: acquired after waiting {time.perf_counter() - tw:.3f} s")
            if problems: raise problems[0]
            return rot[i % 3]

        def release(i):
            if trace: progress(f"  position {i}: released")
            free[i % 3].release()

        th = threading.Thread(target=loader, daemon=True)      # a queue that raised must not leave the process waiting for its loader
        ctx.synchronize()
        tq = time.perf_counter()
        th.start()
        try:
            calls = api.solve_queue_fed(k, acquire, release, stages, master_seeds=[3000 + first_it + i for i in range(k)])
        except BaseException:
            for f in free: f.release()                          # the loader may sit in an acquire: let it run out
            raise
        ctx.synchronize()
        dt = time.perf_counter() - tq
        th.join()
        ok = all(tuple(int(x) for x in gts[int(c.ixs[0])]) == tuple(host[(first_it + i) % D][0].true_genotype) for i, c in enumerate(calls))
        return dt, ok, load_s

    loaded_once = [False]
    run(3, 0)                                                 # every batch object once: workspaces, page tables
    loaded_once[0] = True
    ctx.timing_reset()
    dt, ok, load_s = run(K, 0 if args.distinct_no_upload else 1)
    kern = {name: ctx.timing(k)[1] / K for name, k in (("score_reads_kernel", api.K_SCORE), ("prefilter_tile_kernel", api.K_PREFILTER),
            ("solve_init_kernel", api.K_SOLVE_INIT), ("greedy_loop_kernel", api.K_SOLVE), ("anneal_loop_kernel", api.K_ANNEAL),
            ("build_loc_table_kernel", api.K_SOLVE_TABLE))}
    for b in rot: b.close()
    ms = 1e3 * dt / K
    return {"what": f"{K} positions over {D} distinct loci of {args.pairs} read pairs x {A} alleles; every position uploaded from page-locked host memory "
                    "(lcty_reads_reset + lcty_reads_append_counted from a loader thread, copy stream) while the position before it is solved "
                    "(lcty_solve_queue_fed, three batch objects); the first upload of the queue is inside the timed region",
            "ms_per_step": ms, "read_pairs_per_s": args.pairs * K / dt, "resident_ms_per_step": resident_ms_per_step,
            "ratio_to_resident": ms / resident_ms_per_step, "all_calls_equal_truth": ok, "kernel_ms_per_step": kern,
            "upload_GB_per_locus": up_bytes / 1e9, "upload_and_validate_s_per_locus": float(np.median(load_s)),
            "upload_GBs": up_bytes / 1e9 / float(np.median(load_s)), "setup_s": setup_s}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)           # never returns; nothing above this line loads the HIP library
    # stdout carries exactly one JSON line: whatever the libraries underneath print there (gloo announces its connections, RCCL its
    # version) goes to stderr, the line itself to the real stdout at the very end
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if args.diag: _lib.use_diag_build()
    _lib.lib()      # load the HIP library before anything else can bring another HIP runtime into scope
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist      # gloo: barrier + max-reduce only, no GPU tensors
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if args.gpus != world:
        raise RuntimeError(f"--gpus {args.gpus} but WORLD_SIZE is {world}: the launcher and the command line disagree about the number of ranks")

    ndev = api.device_count()
    if ndev < 1:
        raise RuntimeError("bench.py needs a HIP device (no CPU fallback)")
    if world > ndev and not args.oversubscribe:
        raise RuntimeError(f"{world} ranks but {ndev} HIP device(s) visible: one process per GPU (--oversubscribe puts several ranks on a device "
                           "to exercise the launch path on a small box; the line then says so)")
    ctx = api.Context(local_rank % ndev)
    if world > 1:                                  # this rank's share of the host's cores (spawn_ranks sets it; under torchrun: cores / ranks)
        ctx.set_knob("host_threads", max(1, min(16, int(os.environ.get("LCTY_BENCH_HOST_THREADS", physical_cores() // world)))))
    early_head = True                              # lcty_solve_queue's default (knob queue_early_head)
    for kv in args.knob:
        name, _, val = kv.partition("=")
        ctx.set_knob(name, int(val))
        if name == "queue_early_head": early_head = int(val) != 0

    comm = None
    rccl_ranks = None
    first_pair = 0
    total_pairs = args.pairs
    one_locus = args.shard_reads or args.shard_chains
    if one_locus:
        os.environ.pop("NCCL_DEBUG", None)         # RCCL logs to stdout, which carries the one JSON line
        os.environ["NCCL_DEBUG_FILE"] = os.devnull
        args.recovery_sample = args.recruit_sample = args.ont_sample = args.ont_stream_sample = 0
        uid = api.comm_unique_id() if rank == 0 else bytes(api.COMM_ID_BYTES)
        if dist is not None:
            import torch
            t_uid = torch.tensor(list(uid), dtype=torch.uint8)
            dist.broadcast(t_uid, src=0)
            uid = bytes(t_uid.tolist())
        comm = api.Comm(ctx, world, rank, uid)
        rccl_ranks = comm.rccl_ranks()[0]
    if args.shard_reads:
        per = (args.pairs + world - 1) // world
        first_pair = min(rank * per, args.pairs)
        args.pairs = min(first_pair + per, total_pairs) - first_pair          # this rank's shard

    # ---- synthetic loci + reads (seed + locus index) -> HBM ----
    # The default mode runs a QUEUE of loci through lcty_solve_queue (the loop of `locityper genotype` over its loci): the library
    # overlaps the last stage of a locus (annealing) with the scoring / prefilter / greedy stage of the next one, which needs two
    # loci resident; a step = one locus through the whole path, K steps = a queue of K loci alternating between the two.
    progress("generating the loci and their read pairs")
    t0 = time.time()
    n_loci = 1 if (one_locus or args.no_solve) else 2
    A = args.alleles
    loci, batches, host_chunks = [], [], []
    locus_setup_s = 0.0
    tot_recs = tot_cigar = tot_bases = 0
    first = None
    n_chunks = (args.pairs + args.chunk - 1) // args.chunk
    for j in range(n_loci):
        seed_off = int(args.loci_seeds.split(",")[j]) if args.loci_seeds else (0 if one_locus else n_loci * rank + j)
        L = synth.SynthLocus(A, total_pairs, seed=synth.SEED + seed_off)
        params = api.resolve_params(api.default_params(), L.bg)
        t1 = time.time()
        loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, params)
        ctx.synchronize()
        locus_setup_s += time.time() - t1
        # generate chunk by chunk, keep totals, and allocate from the first chunk's density with head-room
        c0 = L.reads(first_pair, min(args.chunk, args.pairs))
        dens_b, dens_r, dens_c = c0.n_bases / c0.n_pairs, len(c0.recs) / c0.n_pairs, len(c0.cigar) / c0.n_pairs
        head = 1.03
        cap_bases = (int(dens_b * args.pairs * head) + 1024) // 32 * 32 + 32
        counted = args.format == "counted"
        aa = api.AllAlignments(loc, args.pairs, cap_bases, int(dens_r * args.pairs * head) + 4096,
                               0 if counted else int(dens_c * args.pairs * head) + 65536)
        # the chunks of the resident loci are kept on the host when the queue of NON-resident loci is measured afterwards (it uploads them again)
        keep_host = args.distinct_loci >= 2 and not one_locus and not args.no_solve and world == 1 and counted
        host_chunks.append([c0] if keep_host else None)
        aa.append(c0, counted=counted)
        if j == 0:
            tot_recs, tot_cigar, tot_bases = len(c0.recs), len(c0.cigar), c0.n_bases
            first = c0 if (rank == 0 and world == 1 and args.cpu_sample > 0) else None
        for ci in range(1, n_chunks):
            lo = ci * args.chunk
            ch = L.reads(first_pair + lo, min(args.chunk, args.pairs - lo))
            aa.append(ch, counted=counted)
            if j == 0:
                tot_recs += len(ch.recs); tot_cigar += len(ch.cigar); tot_bases += ch.n_bases
            if keep_host: host_chunks[j].append(ch)
            del ch
        loci.append((L, loc)); batches.append(aa)
    L, loc = loci[0]
    aa = batches[0]
    gen_s = time.time() - t0
    G = api.count_genotypes(A, 2)
    all_ixs = np.arange(G, dtype=np.uint64)

    gts = api.generate_genotypes(A, 2)
    greedy, anneal = api.default_solver(cdefs.SOLVER_GREEDY), api.default_solver(cdefs.SOLVER_ANNEAL)
    stage_s = {"score_prefilter": 0.0, "greedy": 0.0, "anneal": 0.0}
    solved = {"greedy_chains": 0, "anneal_chains": 0, "greedy_iterations": 0, "anneal_moves": 0}
    stages = api.default_stages()

    aa_main = aa

    def step(it=0, aa=aa, stage_s=stage_s, solved=solved):
        """One locus through the path, call by call (per-stage wall times; the modes that shard one locus over the ranks)."""
        t0s = time.perf_counter()
        aa.score()
        aa.prefilter_async()
        if args.shard_reads and aa is aa_main:
            comm.prefilter_allreduce(aa)                                      # read shards -> scores of the whole batch on every rank
        scores = None                                                         # they stay on the device: only the kept indices come back
        keep = aa.prefilter_truncate(params.filt_diff, 5000, params.threads)  # truncate_ixs, in_size of stage 1 (solve.rs:216-221)
        stage_s["score_prefilter"] += time.perf_counter() - t0s
        if args.no_solve:
            return scores, keep, None
        # default scheme "-S greedy:i=5k,a=1 -S anneal:i=20,a=20" (solve.rs:211-230), then the final comparison
        n = len(gts)
        mean, var, att = np.full(n, np.nan), np.full(n, np.nan), np.zeros(n, dtype=np.uint32)
        ixs = keep
        # --shard-chains: the same call on every rank, the chains dealt to the ranks inside the library
        # --shard-reads: every rank holds its shard of the locus' reads; a stage exchanges the location-table rows of its alleles
        # (lcty_solve_stage_read_sharded: RCCL all-gathers), then deals its chains to the ranks like --shard-chains
        run_stage = (comm.solve_stage if args.shard_chains else comm.solve_stage_read_sharded) if (one_locus and aa is aa_main) else api.solve_stage
        ts = time.perf_counter()
        if 20 < len(ixs):
            m, v, _ = run_stage(aa, gts[ixs], greedy, 1, api.chain_seeds(1000 + it, len(ixs)))
            mean[ixs], var[ixs], att[ixs] = m, v, 1
            solved["greedy_chains"] += len(ixs)
            solved["greedy_iterations"] += api.solve_stats(aa)[1]
            ixs = api.discard_improbable(mean, var, att, ixs, params.prob_thresh, 20, params.threads)
        tm = time.perf_counter()
        m, v, _ = run_stage(aa, gts[ixs], anneal, 20, api.chain_seeds(2000 + it, 20 * len(ixs)))
        mean[ixs], var[ixs], att[ixs] = m, v, 20
        solved["anneal_chains"] += 20 * len(ixs)
        solved["anneal_moves"] += api.solve_stats(aa)[1]
        res = api.produce_result(mean, var, att, ixs, params.prob_thresh)
        te = time.perf_counter()
        stage_s["greedy"] += tm - ts; stage_s["anneal"] += te - tm
        return scores, keep, res

    queue_mode = n_loci == 2

    def run_steps(k, first_it=0):
        """k steps: a queue of k loci through lcty_solve_queue (default), or k passes call by call."""
        if queue_mode:
            order = [(first_it + i) % 2 for i in range(k)]
            return api.solve_queue([batches[j] for j in order], stages, master_seeds=[1000 + first_it + i for i in range(k)]), order
        out = None
        for i in range(k):
            out = step(first_it + i)
        return out, None

    def barrier():
        ctx.synchronize()
        if dist is not None:
            dist.barrier()

    if args.warmup > 0:
        run_steps(max(args.warmup, 2) if queue_mode else args.warmup)        # both loci once: allocations, the solver workspaces
    barrier()
    ctx.timing_reset()
    stage_s.update(score_prefilter=0.0, greedy=0.0, anneal=0.0); solved.update(greedy_chains=0, anneal_chains=0, greedy_iterations=0, anneal_moves=0)
    t_start = time.perf_counter()
    result, order = run_steps(args.steps, 100)
    ctx.synchronize()
    elapsed = time.perf_counter() - t_start
    barrier()
    rank_ms = None
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        lo = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        # the step of the line is the slowest rank's; the fastest beside it shows how evenly the ranks ran
        rank_ms = {"min": 1e3 * float(lo[0]) / args.steps, "max": 1e3 * float(t[0]) / args.steps}
        elapsed = float(t[0])
    n_score, ms_score = ctx.timing(api.K_SCORE)
    n_pref, ms_pref = ctx.timing(api.K_PREFILTER)
    n_solve, ms_solve = ctx.timing(api.K_SOLVE)
    n_ann, ms_ann = ctx.timing(api.K_ANNEAL)
    n_init, ms_init = ctx.timing(api.K_SOLVE_INIT)
    n_init_a, ms_init_a = ctx.timing(api.K_SOLVE_INIT_ANNEAL)
    n_tab, ms_tab = ctx.timing(api.K_SOLVE_TABLE)
    calls_ok = None
    if queue_mode:
        calls = result
        called = tuple(int(x) for x in gts[int(calls[-1].ixs[0])])
        truth = loci[order[-1]][0].true_genotype
        calls_ok = all(tuple(int(x) for x in gts[int(c.ixs[0])]) == tuple(loci[j][0].true_genotype) for c, j in zip(calls, order))
        kept = int(calls[-1].kept_after_filter)
        quality = float(calls[-1].quality)
        greedy_chains_per_step = float(np.mean([int(c.kept_after_filter) for c in calls]))
        # per-stage wall times and iteration counts: one more locus call by call, outside the timed region
        alone_ms = {}
        if rank == 0:
            ctx.timing_reset()                                  # the queue's timers have been read: what follows are the kernels with nothing beside them
            step(7)
            ctx.synchronize()
            for name, kid in (("score_reads_kernel", api.K_SCORE), ("prefilter_tile_kernel", api.K_PREFILTER), ("solve_init_kernel", api.K_SOLVE_INIT),
                              ("solve_init_kernel_annealing_stage", api.K_SOLVE_INIT_ANNEAL), ("greedy_loop_kernel", api.K_SOLVE), ("anneal_loop_kernel", api.K_ANNEAL)):
                nk, msk = ctx.timing(kid)
                if nk: alone_ms[name] = msk / nk
        res = None
    else:
        scores, keep, res = result
        top = int(keep[0]) if res is None else int(res[0][0])
        called = tuple(int(x) for x in gts[top]); truth = L.true_genotype
        kept = int(len(keep)); quality = None if res is None else float(res[2])
        greedy_chains_per_step = solved["greedy_chains"] / max(args.steps, 1)

    if rank != 0:
        return

    n_break = 1 if queue_mode else args.steps                  # steps behind stage_s / solved
    ms_per_step = 1e3 * elapsed / args.steps
    reads_per_s = (total_pairs if one_locus else world * args.pairs) * args.steps / elapsed
    score_ms = ms_score / max(n_score, 1)
    pref_ms = ms_pref / max(n_pref, 1)
    kern_steps = args.steps                                    # the timers were read before the call-by-call pass
    alg_bytes = survey_bytes_per_pair(A) * args.pairs
    layout_bytes = (tot_bases / 4 + tot_bases / 8 + 16 * tot_recs + (0 if args.format == "counted" else 4 * tot_cigar) + 8 * A * args.pairs
                    + 8 * 4 * args.pairs)      # what the kernel's inputs/outputs occupy, excl. pair-alignment arena
    n_good = aa.n_good()
    # ---- rooflines of the kernels of a step; `roofline` is the one with the most kernel time (DESIGN.md section 4 for the bytes) ----
    per_step = {k: v / max(n_break, 1) for k, v in solved.items()}
    chains_step = per_step["greedy_chains"] + per_step["anneal_chains"]
    roofs = {
        "score_reads_kernel": {"ms_per_step": ms_score / kern_steps, "launch_ms": score_ms, "bound": "hbm",
                               "bytes": alg_bytes, "what": "SURVEY 8(d): 75 + 2*A*16 + 2016 + A*8 B per read pair"},
        "prefilter_tile_kernel": {"ms_per_step": ms_pref / kern_steps, "launch_ms": pref_ms, "bound": "valu_f64",
                                  "ops": 2.0 * G * args.pairs, "what": "2 * G * R max-add"},
        # the two initialisations of a step apart: the greedy stage's chains (~5 000, main stream) and the annealing stage's (400, side stream)
        "solve_init_kernel": {"ms_per_step": ms_init / kern_steps, "launch_ms": ms_init / max(n_init, 1), "bound": "hbm",
                              "bytes": (34.0 * n_good + 207e3) * per_step["greedy_chains"] if chains_step else 0.0,
                              "chains_per_launch": per_step["greedy_chains"],
                              "what": "solve_init_tile_kernel on the chains of the greedy stage; SURVEY 8(d): the reads CSR once per genotype x attempt, "
                                      "34 B * R + 207 KB LUT per chain (the layout: 32 B written + 16 B x rows / chains read per chain and read)"},
        "solve_init_kernel_annealing_stage": {"ms_per_step": ms_init_a / kern_steps, "launch_ms": ms_init_a / max(n_init_a, 1), "bound": "hbm",
                              "bytes": (34.0 * n_good + 207e3) * per_step["anneal_chains"] if chains_step else 0.0,
                              "chains_per_launch": per_step["anneal_chains"],
                              "what": "the same kernel on the chains of the annealing stage (side stream, beside the next locus)"},
        "greedy_loop_kernel": {"ms_per_step": ms_solve / kern_steps, "launch_ms": ms_solve / max(n_solve, 1), "bound": "hbm",
                               "bytes": 32.0 * 10 * per_step["greedy_iterations"],
                               "what": "one 32 B record per candidate read, 10 candidates per iteration"},
        "anneal_loop_kernel": {"ms_per_step": ms_ann / kern_steps, "launch_ms": ms_ann / max(n_ann, 1), "bound": "hbm",
                               "bytes": 32.0 * per_step["anneal_moves"], "what": "one 32 B record per evaluated move (latency-bound serial chains)"},
    }
    if queue_mode and early_head:
        # lcty_solve_queue issues everything before the chains of a locus on a third stream, beside the greedy chains of the locus before
        # (one wavefront per SIMD, 125 of 160 KB of LDS): launch_ms of these two is the kernel in THAT place — off the critical path of a step —
        # and launch_ms_alone / frac_alone the kernel with the device to itself
        for name in ("score_reads_kernel", "prefilter_tile_kernel"):
            roofs[name]["in_the_queue"] = "fore stream, beside the greedy chains of the locus before; not on the critical path of a step"
    # the loop kernels are random 32-byte gathers out of the chains' 148 GB of records: what the device does of THOSE at best (a lane keeps
    # one to four in flight, 1 250 - 10 000 wavefronts: 37.9 G gathers/s, profiles/r05_gather_probe.txt) is the ceiling their record
    # gathers are held against; SURVEY 8(d) itself calls K14 latency-bound
    GATHER_CEILING = 37.9e9
    for name, gathers in (("greedy_loop_kernel", 10.0 * per_step["greedy_iterations"]), ("anneal_loop_kernel", per_step["anneal_moves"])):
        r = roofs[name]
        if r["launch_ms"] > 0:
            r["record_gathers_per_s"] = gathers / (r["launch_ms"] * 1e-3)
            r["gather_ceiling_frac"] = r["record_gathers_per_s"] / GATHER_CEILING
            r["gather_ceiling"] = {"gathers_per_s": GATHER_CEILING, "source": "profiles/r05_gather_probe.txt (scripts/gather_probe.hip: 148 GB footprint)"}
    for k, r in roofs.items():
        if r["bound"] == "hbm":
            r["achieved"] = r["bytes"] / max(r["launch_ms"], 1e-9) / 1e6; r["peak"] = HBM_PEAK_GBS; r["unit"] = "GB/s"
        else:
            r["achieved"] = r["ops"] / max(r["launch_ms"], 1e-9) / 1e9; r["peak"] = 39.3; r["unit"] = "Tmaxadd/s"
        r["frac"] = r["achieved"] / r["peak"]
    # the dominant kernel of a step: the one with the most time on the MAIN stream, whose kernels run back to back and make up the
    # step; the annealing chains of the locus before run on the side stream next to them (overlapped, never on the critical path of the
    # queue) and are reported in roofline_all like everything else
    dominant = max((k for k in roofs if roofs[k]["bound"] == "hbm" and k != "anneal_loop_kernel"), key=lambda k: roofs[k]["ms_per_step"])
    achieved = roofs["score_reads_kernel"]["achieved"]
    out = {
        "metric": "reads/s through the whole genotyping path (scored + prefiltered + default solver scheme); "
                  "reads_scored_per_s and genotypes_solved_per_s give the two halves",
        "value": reads_per_s,
        "unit": "read pairs/s",
        "n_gpus": world,
        "devices_used": min(world, ndev),
        "launch": os.environ.get("LCTY_BENCH_LAUNCH", "launcher environment (torchrun)" if world > 1 else "single process"),
        "rccl_ranks": rccl_ranks,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "ms_per_step_ranks": rank_ms,
        "higher_is_better": True,
        "scaling": "strong" if one_locus else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"{args.pairs} synthetic 150 bp PE read pairs x {A} alleles, 1 locus per step, k=25 "
                               + ("(BASELINE.json configs[1])" if (total_pairs, A) == (1_000_000, 256) and not one_locus
                                  else "(one GPU's share of BASELINE.json configs[4])" if A == 4096
                                  else "(not a BASELINE.json configuration)"),
                   "read_pairs": args.pairs, "alleles": A, "genotypes": G, "k": 25,
                   "records": tot_recs, "cigar_words": tot_cigar,
                   "alignment_table": ("16-byte counted alignments (lcty_reads_append_counted): the caller counts the CIGAR operations"
                                       if args.format == "counted" else "16-byte BAM records + CIGAR words (lcty_reads_append)"),
                   "step": ("one locus through lcty_solve_queue (score + run_filter + default solver scheme + final comparison); the queue "
                            "alternates between two resident loci; beside the greedy chains of a locus run the annealing stage of the locus "
                            "before (side stream) and" + (" the scores, run_filter and location table of the locus after (fore stream)" if early_head else " nothing else")
                            if queue_mode else "one locus, call by call"),
                   "parallelism": (f"reads of one locus x{world}: RCCL all-reduce of the run_filter scores, all-gather of the location-table rows per solver stage, chains dealt to the ranks" if args.shard_reads else f"solver chains of one locus x{world} (reads replicated), RCCL all-gather of the chain likelihoods" if args.shard_chains else f"loci x{world}")},
        "reads_scored_per_s": (total_pairs if args.shard_reads else args.pairs if args.shard_chains else args.pairs) * n_break / max(stage_s["score_prefilter"], 1e-9),
        "genotypes_prefiltered_per_s": G * n_break / max(stage_s["score_prefilter"], 1e-9),
        "prefilter_genotypes_per_s_kernel": G / (pref_ms * 1e-3) if pref_ms else None,
        "kernel_ms_per_step": {k: r["ms_per_step"] for k, r in roofs.items()} | {"build_loc_table_kernel": ms_tab / kern_steps},
        "solver": None if args.no_solve else {
            "scheme": "greedy:i=5k,a=1 -> anneal:i=20,a=20 -> final comparison",
            "genotypes_solved_per_s": world * (greedy_chains_per_step + 20) * args.steps / elapsed,
            "chains_per_s": world * (greedy_chains_per_step + 400) * args.steps / elapsed,
            "per_step": per_step,
            "call_by_call_stage_ms": {k: 1e3 * v / max(n_break, 1) for k, v in stage_s.items()},
            "quality": quality, "all_calls_equal_truth": calls_ok},
        "roofline": {"bound": "hbm", "kernel": dominant, "achieved": roofs[dominant]["achieved"], "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": roofs[dominant]["frac"], "traffic": None,
                     "algorithmic_bytes_per_launch": roofs[dominant]["bytes"], "launch_ms": roofs[dominant]["launch_ms"],
                     "what": roofs[dominant]["what"]},
        "roofline_all": roofs,
        "roofline_score_layout": {"layout_bytes_per_launch": layout_bytes, "achieved_layout_GBs": layout_bytes / (score_ms * 1e-3) / 1e9 if score_ms else None},
        "kernel_sources_sha16": sources_sha16(ROOT),
        "called_genotype": called, "true_genotype": truth, "kept_after_prefilter": kept,
        "setup_s": {"generate_and_upload": gen_s, "locus_create": locus_setup_s},
    }

    # the same kernels of one more locus solved call by call, nothing else on the device (in the queue the last stage of the locus before runs beside them)
    if queue_mode:
        for name, r in roofs.items():
            if name in alone_ms and alone_ms[name] > 0:
                r["launch_ms_alone"] = alone_ms[name]
                if "bytes" in r: r["frac_alone"] = r["bytes"] / (alone_ms[name] * 1e-3) / 1e9 / 8000.0
    # HBM traffic from the committed PMC passes (counters cannot be read from inside this process). FETCH_SIZE on gfx950 counts a 128-byte
    # read request as 64 bytes (MI355X_MICROARCH.md): doubled for the kernels that STREAM wide coalesced reads; kernels that gather 8-32
    # bytes per lane are outside that calibration and keep the raw figure. Both are in the line.
    streaming = {"score_reads_kernel", "solve_init_kernel", "prefilter_tile_kernel"}
    try:
        tr = json.load(open(args.traffic))
        if tr.get("read_pairs") == args.pairs and tr.get("alleles") == A:
            for name, r in roofs.items():
                cands = [v for n, v in tr["kernels"].items() if name.replace("score_reads_kernel", "score_") in n or name.replace("solve_init_kernel", "solve_init_tile_kernel") in n]
                k = max(cands, key=lambda v: v.get("fetch_bytes_raw", 0.0) + v.get("write_bytes", 0.0)) if cands else None
                if k:
                    r["traffic_fetch_raw"] = k.get("fetch_bytes_raw"); r["traffic_fetch_x2"] = 2.0 * k.get("fetch_bytes_raw", 0.0); r["traffic_write"] = k.get("write_bytes")
                    r["traffic_rule"] = "2 x FETCH_SIZE + WRITE_SIZE (streaming reads)" if name in streaming else "FETCH_SIZE + WRITE_SIZE (narrow gathers: raw)"
                    r["traffic"] = (2.0 if name in streaming else 1.0) * k.get("fetch_bytes_raw", 0.0) + k.get("write_bytes", 0.0)
            for key in ("traffic", "traffic_fetch_raw", "traffic_fetch_x2", "traffic_write", "traffic_rule"):
                out["roofline"][key] = roofs[dominant].get(key)
            out["roofline"]["traffic_source"] = os.path.relpath(args.traffic, ROOT)
            # a counter file from other kernels than the ones timed here must show: the library's sources are hashed next to every pass
            out["roofline"]["traffic_taken_at_commit"] = tr.get("taken_at_commit")
            out["roofline"]["traffic_sources_sha16"] = tr.get("sources_sha16")
            out["roofline"]["traffic_is_current"] = tr.get("sources_sha16") == out["kernel_sources_sha16"]
    except (OSError, KeyError, ValueError):
        pass
    # what the wavefronts of each kernel were doing (committed SQ counter pass, scripts/pmc_sq_summary.py): the fraction of their cycles with
    # an instruction in flight / waiting. issuing x wavefronts per SIMD near or above 1 = the kernel is bound by its instruction stream.
    try:
        sq_doc = json.load(open(os.path.join(ROOT, "profiles", "r05_pmc_sq.json")))
        sq = sq_doc["kernels"]
        per_simd = {"greedy_loop_kernel": 1.0, "solve_init_kernel": 2.0, "score_reads_kernel": 4.0, "anneal_loop_kernel": 0.8, "prefilter_tile_kernel": None}
        for name, r in roofs.items():
            cands = [v for n, v in sq.items() if name.replace("score_reads_kernel", "score_counted_lean").replace("solve_init_kernel", "solve_init_tile_kernel") in n]
            if cands:
                k = max(cands, key=lambda v: v.get("SQ_WAVE_CYCLES", 0.0))
                r["sq"] = {"issuing_frac": round(k.get("active_inst_frac", 0.0), 3), "waiting_frac": round(k.get("wait_any_frac", 0.0), 3),
                           "wavefronts_per_simd": per_simd.get(name), "source": "profiles/r05_pmc_sq.json",
                           "is_current": sq_doc.get("sources_sha16") == out["kernel_sources_sha16"]}
                # vector instructions issued per launch against what the SIMDs could issue in the launch's time (one per 4 cycles each: quarter-
                # and half-rate instructions take longer, so a kernel is bound by its vector stream well below 1)
                launches = max(k.get("launches", 0.0), 1.0)
                # (a kernel that runs beside the greedy chains in the queue: against its time with the device to itself)
                ms_for_issue = r.get("launch_ms_alone") if r.get("in_the_queue") and r.get("launch_ms_alone") else r.get("launch_ms")
                if k.get("SQ_INSTS_VALU") and ms_for_issue:
                    r["sq"]["valu_issue_frac"] = (k["SQ_INSTS_VALU"] / launches) / (1024 * 2.4e9 / 4.0 * ms_for_issue * 1e-3)
        out["roofline"]["sq"] = roofs[dominant].get("sq")
    except (OSError, KeyError, ValueError):
        pass

    if world == 1:
        progress("timed region done")
        ctx.trim()          # the solver workspaces of the timed steps (160 GB) make room for the extra measurements below

    if args.recruit_sample > 0 and world == 1:
        progress("recruitment leg")
        # ---- minimizer read recruitment (Targets::recruit_read_pair, seq/recruit.rs:883-929), the step before the path: random
        # 150 + 150-base pairs (whole-genome input is almost entirely foreign to a locus) against this locus' alleles ----
        nrq = args.recruit_sample
        rprm = api.recruit_params()
        tq0 = time.perf_counter()
        T = api.Targets(ctx, rprm)
        T.add_locus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k)
        n_minim = T.finalize()
        t_targets = time.perf_counter() - tq0
        rngq = np.random.default_rng(11)
        words = rngq.integers(0, 1 << 32, size=nrq * 20, dtype=np.uint64).astype(np.uint32)
        # 0.2 % of the pairs come from the locus itself (the share of a whole-genome sample a 50-kb locus accounts for is far smaller;
        # this is what makes the positive path run): their bases over the random ones, at random places of the chunk
        n_own = max(1, nrq // 500)
        own = L.reads(0, min(n_own, args.pairs))
        n_own = own.n_pairs
        own_at = rngq.choice(nrq, size=n_own, replace=False)
        mo = own.mate_off.astype(np.int64) // 16
        short = 0
        for t in range(n_own):
            for e in range(2):
                if int(own.mate_len[2 * t + e]) != 150: short += 1; continue
                words[(2 * int(own_at[t]) + e) * 10:(2 * int(own_at[t]) + e) * 10 + 10] = own.bases2[mo[2 * t + e]:mo[2 * t + e] + 10]
        from locityper_amd.cdefs import ReadsChunk, ALN_REC_DTYPE
        rq = ReadsChunk(np.full(2 * nrq, 150, dtype=np.uint32), np.arange(2 * nrq + 1, dtype=np.uint64) * 160, words,
                        np.zeros(nrq * 10, dtype=np.uint32), np.zeros(nrq + 1, dtype=np.uint64), np.zeros(0, dtype=ALN_REC_DTYPE),
                        np.zeros(nrq + 1, dtype=np.uint64), np.zeros(0, dtype=np.uint32))
        ctx.timing_reset()
        tq0 = time.perf_counter()
        cntq, _ = T.recruit(rq, paired=True)
        t_call = time.perf_counter() - tq0
        _, ms_q = ctx.timing(api.K_RECRUIT)
        out["recruitment"] = {"sample": f"{nrq} 150 + 150-base read pairs, 99.8 % random and 0.2 % drawn from the locus, against its {A} alleles ({n_minim} minimizers)",
                              "kernel_ms": ms_q, "read_pairs_per_s_kernel": nrq / (ms_q * 1e-3) if ms_q else None,
                              "read_pairs_per_s_call": nrq / t_call, "targets_build_s": t_targets, "recruited": int(np.count_nonzero(cntq)),
                              "locus_derived_pairs": int(n_own), "locus_derived_recruited": int(np.count_nonzero(cntq[own_at]))}
        if n_own - short // 2 > 10 and out["recruitment"]["locus_derived_recruited"] == 0:
            raise RuntimeError("recruitment leg: none of the locus' own read pairs was recruited")
        T.close(); del rq, words

    if args.map_sample > 0 and world == 1:
        progress("candidate-generation leg")
        # ---- candidate generation on the basis alleles (first slice; lcty_map.hip), the step the reference leaves to
        # an external mapper: the first read pairs of the locus, their bases only, onto 8 basis alleles ----
        nmp = min(args.map_sample, args.pairs)
        src = L.reads(0, nmp)
        from locityper_amd.cdefs import ReadsChunk, ALN_REC_DTYPE
        zoff = np.zeros(nmp + 1, dtype=np.uint64)
        bare = ReadsChunk(src.mate_len, src.mate_off, src.bases2, src.nmask, zoff, np.zeros(0, dtype=ALN_REC_DTYPE), zoff, np.zeros(0, dtype=np.uint32))
        mp = api.map_params()
        basis = list(range(0, A, max(1, A // 8)))[:8]
        tm0 = time.perf_counter()
        api.build_map_index(loc, basis, k=mp.k)
        t_index = time.perf_counter() - tm0
        tm0 = time.perf_counter()
        mapped = api.map_reads(loc, bare, mp)                                     # to the host: sizes, then records (the chunk is mapped twice); warm-up
        t_host = time.perf_counter() - tm0
        # the path of a run: the records straight into a batch of the locus (device to device), one mapping
        am = api.AllAlignments(loc, nmp, (int(src.n_bases) + 2048) // 32 * 32, len(mapped.recs) + 1024, len(mapped.cigar) + 1024)
        ctx.timing_reset()
        tm0 = time.perf_counter()
        api.map_append(am, bare, mp)
        t_map = time.perf_counter() - tm0
        _, ms_map = ctx.timing(api.K_MAP)
        am.close()
        out["candidate_generation"] = {"sample": f"first {nmp} read pairs (bases only) onto {len(basis)} basis alleles, seeds of {mp.k} every {mp.stride} bases, ungapped extension then a band alignment with gaps for clipped candidates; records straight into a batch (lcty_reads_map_append)",
                                       "records": int(len(mapped.recs)), "kernel_ms": ms_map,
                                       "read_ends_per_s_kernel": 2 * nmp / (ms_map * 1e-3) if ms_map else None,
                                       "read_ends_per_s_call": 2 * nmp / t_map, "to_host_two_mappings_s": t_host, "index_build_s": t_index}
        del src, bare, mapped

    if args.recovery_sample > 0 and world == 1:
        progress("alignment-recovery leg")
        # ---- alignment recovery (K6), not part of the step: the mapper reports only the primary alignment of each read end, the
        # other alleles are reached through the haplotype-to-haplotype alignments (transfer.rs:70-140) ----
        nrec = min(args.recovery_sample, args.pairs)
        tr0 = time.perf_counter()
        H = L.hap_alns()
        loc.set_hap_alns(H, transfer_fails=100, max_div=0.1)                   # genotype.rs:149-150 defaults
        t_set = time.perf_counter() - tr0
        prim = [L.reads(lo, min(args.chunk, nrec - lo), primaries_only=True) for lo in range(0, nrec, args.chunk)]
        ab = api.AllAlignments.load(loc, prim)
        ctx.timing_reset()
        tr0 = time.perf_counter()
        n_new = ab.recover()
        t_rec = time.perf_counter() - tr0
        n_tr, ms_tr = ctx.timing(5)                                            # LCTY_K_TRANSFER
        cells_sr = ab.recover_dp_cells()
        out["recovery"] = {"aligner_cells": int(cells_sr), "gcups": cells_sr / (ms_tr * 1e-3) / 1e9 if ms_tr else None,"sample": f"first {nrec} read pairs, primary records only ({sum(len(c.recs) for c in prim)} records), "
                                     f"{len(H)} haplotype alignments, transfer_fails 100",
                           "alignments_transferred": int(n_new), "transfer_kernel_ms": ms_tr, "launches": int(n_tr),
                           "transfers_per_s": n_new / (ms_tr * 1e-3) if ms_tr else None,
                           "recover_and_rescore_s": t_rec, "set_hap_alns_s": t_set, "good_pairs_after": ab.n_good(),
                           "level_pairs": ab.recover_stats()}
        ab.close(); del prim

    if args.many_alleles_sample > 0 and world == 1:
        progress("many-alleles leg")
        # ---- configs[4] shard shape: a locus of 4 096 alleles, the read pairs one of eight GPUs would hold. The prefilter is the
        # dominant kernel there; from 512 alleles on it runs as an integer Gram contraction on the matrix cores (lcty_gram.hip) ----
        nma, Ama = args.many_alleles_sample, 4096
        Lm = synth.SynthLocus(Ama, nma, seed=synth.SEED + 5, base_len=3000)
        pm = api.resolve_params(api.default_params(), Lm.bg)
        locm = api.Locus(ctx, Lm.seqs, Lm.seq_off, Lm.counts, Lm.cnt_off, Lm.k, Lm.bg, pm)
        am = None
        for lo in range(0, nma, 4096):
            chm = Lm.reads(lo, min(4096, nma - lo))
            if am is None:
                fm = 1.1 * nma / chm.n_pairs
                am = api.AllAlignments(locm, nma, (int(chm.n_bases * fm) + 2048) // 32 * 32, int(len(chm.recs) * fm) + 4096, int(len(chm.cigar) * fm) + 65536)
            am.append(chm)
        ctx.timing_reset()
        am.score(); ctx.synchronize()
        _, ms_score_m = ctx.timing(api.K_SCORE)
        leg = {"workload": f"{nma} read pairs x {Ama} alleles ({Ama * (Ama + 1) // 2} genotypes): BASELINE configs[4] gives each of eight GPUs 625 000 read pairs of such a locus",
               "score_reads_kernel_ms": ms_score_m}
        scores_m = {}
        for name, knob in (("f64_tile_kernel", 0), ("integer_gram_on_mfma", 1)):
            ctx.set_knob("prefilter_gram", knob)
            am.prefilter_async(); ctx.synchronize()
            ctx.timing_reset()
            am.prefilter_async(); ctx.synchronize()
            leg[name + "_ms"] = ctx.timing(api.K_PREFILTER)[1]
            scores_m[name] = am.prefilter_scores()
        ctx.set_knob("prefilter_gram", -1)
        leg["max_relative_difference"] = float(np.abs(scores_m["integer_gram_on_mfma"] - scores_m["f64_tile_kernel"]).max() / np.abs(scores_m["f64_tile_kernel"]).max())
        best_m = api.generate_genotypes(Ama, 2)[int(np.argmax(scores_m["integer_gram_on_mfma"]))]
        leg["best_genotype"] = [int(x) for x in best_m]; leg["true_genotype"] = list(Lm.true_genotype)
        out["many_alleles"] = leg
        am.close(); del scores_m

    if args.ont_sample > 0 and world == 1:
        progress("long-read leg")
        # ---- configs[2] shape, the long-read DP path: 10-kb single-end ONT reads, primaries only, every other allele reached by
        # HapAlns::transfer_alignments (two-CIGAR walk + gap-affine aligner on the stretches between anchors) ----
        nont = args.ont_sample
        Lo = synth.SynthLocus(A, nont, seed=synth.SEED + 77, technology=cdefs.TECH_NANOPORE, read_len=10_000)
        po = api.resolve_params(api.default_params(), Lo.bg)
        loco = api.Locus(ctx, Lo.seqs, Lo.seq_off, Lo.counts, Lo.cnt_off, Lo.k, Lo.bg, po)
        tr0 = time.perf_counter()
        Ho = Lo.hap_alns()
        loco.set_hap_alns(Ho, transfer_fails=100, max_div=0.1)
        t_set = time.perf_counter() - tr0
        chunk_o = 256
        prim = [Lo.reads(lo, min(chunk_o, nont - lo), primaries_only=True) for lo in range(0, nont, chunk_o)]
        ao = api.AllAlignments.load(loco, prim)
        tr0 = time.perf_counter()
        ao.recover()                                                             # the first call allocates the lane scratch of the context (tens of GB)
        t_rec_first = time.perf_counter() - tr0
        ao.close()
        ao = api.AllAlignments.load(loco, prim)
        ctx.timing_reset()
        tr0 = time.perf_counter()
        n_new = ao.recover()
        t_rec = time.perf_counter() - tr0
        n_tr, ms_tr = ctx.timing(api.K_TRANSFER)
        _, ms_sc = ctx.timing(api.K_SCORE)
        cells = ao.recover_dp_cells()
        out["long_reads"] = {"sample": f"{nont} synthetic 10-kb ONT reads x {A} alleles (BASELINE.json configs[2] shape), primary records only, "
                                       f"{len(Ho)} haplotype alignments, transfer_fails 100",
                             "alignments_transferred": int(n_new), "transfer_kernel_ms": ms_tr, "launches": int(n_tr),
                             "transfers_per_s": n_new / (ms_tr * 1e-3) if ms_tr else None,
                             "aligner_cells": int(cells), "gcups": cells / (ms_tr * 1e-3) / 1e9 if ms_tr else None,
                             "bases_walked_per_s": n_new * 10_000 / (ms_tr * 1e-3) if ms_tr else None,
                             "second_scoring_pass_ms": ms_sc, "recover_and_rescore_s": t_rec, "first_call_s": t_rec_first, "set_hap_alns_s": t_set,
                             "set_hap_alns_library_call_s": loco.set_hap_alns_call_s,
                             "good_reads_after": ao.n_good(), "level_pairs": ao.recover_stats()}
        # the recovered table carries the truth: run_filter over all genotypes on it must put the genotype the reads were drawn from first
        sc_o = ao.run_filter()
        out["long_reads"]["prefilter_best_is_truth"] = bool(tuple(int(x) for x in gts[int(np.argmax(sc_o))]) == tuple(Lo.true_genotype))
        # per transfer the kernel has to look at the read's CIGAR (4 B per item), the part of the haplotype-to-haplotype CIGAR under the read
        # (8 B per item), the target's bases under the read, and write the transferred CIGAR (4 B per item): its algorithmic bytes — a small
        # fraction of the roofline: the walk is bound by instruction issue
        cig_items = float(sum(len(c.cigar) for c in prim)) / max(sum(c.n_pairs for c in prim), 1)
        hap_items = float(np.mean([len(h[2]) for h in Ho[:512]])) * 10_000.0 / float(np.mean(np.diff(Lo.seq_off)))
        per_transfer = 4.0 * cig_items + 8.0 * hap_items + 10_000.0 + 4.0 * cig_items
        out["long_reads"]["roofline"] = {"bound": "hbm", "kernel": "transfer_kernel", "achieved": per_transfer * n_new / (ms_tr * 1e-3) / 1e9 if ms_tr else None,
                                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": per_transfer * n_new / (ms_tr * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_tr else None,
                                         "traffic": None, "algorithmic_bytes_per_transfer": per_transfer,
                                         "what": "read CIGAR + haplotype CIGAR under the read + target bases under the read + transferred CIGAR, per transfer"}
        try:
            # HBM bytes per transfer from the committed counter passes over the same kernel (scripts/run_transfer_traffic.sh), times this leg's transfers
            with open(os.path.join(ROOT, "profiles", "r05_pmc_transfer_kernel.json")) as f:
                tt = json.load(f)
            bpt = tt["bytes_per_transfer"]
            out["long_reads"]["roofline"].update({"traffic": (bpt["fetch_raw"] + bpt["write"]) * n_new, "traffic_bytes_per_transfer": bpt,
                                                   "traffic_rule": "FETCH_SIZE + WRITE_SIZE (narrow gathers: raw)", "traffic_source": "profiles/r05_pmc_transfer_kernel.json",
                                                   "traffic_is_current": tt.get("sources_sha16") == out["kernel_sources_sha16"]})
        except (OSError, KeyError, ValueError):
            pass
        ao.close(); del prim
        if args.ont_map_sample > 0:
            # ---- the same shape from bases alone (second slice; lcty_map_long.hip): the reads as the sequencer gave
            # them are mapped onto EVERY allele (seed - chain - gap-affine alignment along the chain; the reference runs minimap2 here,
            # genotype.rs:990-1002, --basis none), the records go straight into a batch: per-read alignment against every allele on
            # the device, no external mapper, no recovery needed ----
            progress("long-read leg from bases alone")
            ctx.trim()                                                              # the solver workspaces of the timed region (150 GB) are not this leg's business
            nmap = min(args.ont_map_sample, nont)
            chunk_m = 1024                                                          # read ends per mapping call: one wavefront each in the chain kernel
            fq = [synth.sequencer_orientation(Lo.reads(lo, min(chunk_m, nmap - lo), primaries_only=True)) for lo in range(0, nmap, chunk_m)]
            mpl = api.map_params(long_reads=True)
            basis_l = list(range(A))
            tm0 = time.perf_counter()
            api.build_map_index(loco, basis_l, k=mpl.k)
            t_index = time.perf_counter() - tm0
            tot_bases = sum(int(c.n_bases) for c in fq); read_bases = sum(int(c.mate_len.sum()) for c in fq)
            cap_cig = int(read_bases // 3) * len(basis_l) + 4096

            def empty_batch():
                return api.AllAlignments(loco, nmap, (tot_bases + 2048) // 32 * 32, nmap * len(basis_l) * 2 + 1024, cap_cig)
            am = empty_batch()                                                      # warm-up: code objects, and the buffers of the mapping, which stay
            api.map_append(am, fq[0], mpl)                                          # with the batch (tens of GB: the first allocation after the solver
            am.reset(loco)                                                          # workspaces were released takes seconds); then the batch empty again
            ctx.timing_reset()
            tm0 = time.perf_counter()
            for c in fq:
                api.map_append(am, c, mpl)
            t_map = time.perf_counter() - tm0
            n_launch, ms_map = ctx.timing(api.K_MAP)
            # scoring, recovery with its second scoring pass (a first pass looks where a transfer would start at all: the mapper has reached
            # every allele, so there is next to nothing) and run_filter, each on its own clock
            tm0 = time.perf_counter()
            am.score()
            ctx.synchronize()
            t_score1 = time.perf_counter() - tm0
            n_mapped = int(am.pair_alns()[0][-1])
            tm0 = time.perf_counter()
            n_rec2 = am.recover()
            ctx.synchronize()
            t_recover = time.perf_counter() - tm0
            tm0 = time.perf_counter()
            sc_m = am.run_filter()
            ctx.synchronize()
            t_filter = time.perf_counter() - tm0
            t_rest = t_score1 + t_recover + t_filter
            band_w = 2 * mpl.band + 1
            cells_m = float(n_mapped) * (read_bases / max(nmap, 1)) * band_w
            per_aln = read_bases / max(nmap, 1) * (0.25 + 1.0 + 1.0)               # packed read bases + allele bases under the read + ~a CIGAR word per 4 bases
            out["long_reads"]["from_bases"] = {
                "sample": f"the first {nmap} of those reads as sequenced (no records) onto all {len(basis_l)} alleles: seeds of {mpl.k} every {mpl.stride} bases, "
                          f"one chain per (allele, strand), gap-affine alignment along the chain in a band of +-{mpl.band}; records straight into a batch "
                          f"(lcty_reads_map_append, chunks of {chunk_m}), then scoring (+ recovery of the few alignments the mapper left out) + prefilter",
                "alignments": n_mapped, "map_kernels_ms": ms_map, "launches": int(n_launch), "map_call_s": t_map, "index_build_s": t_index,
                "alignments_per_s_kernel": n_mapped / (ms_map * 1e-3) if ms_map else None, "reads_per_s_call": nmap / t_map,
                "aligned_bases_per_s_kernel": n_mapped * (read_bases / max(nmap, 1)) / (ms_map * 1e-3) if ms_map else None,
                "band_cells": cells_m, "gcups": cells_m / (ms_map * 1e-3) / 1e9 if ms_map else None,
                "score_recover_rescore_s": t_rest, "score_s": t_score1, "recover_s": t_recover, "run_filter_s": t_filter,
                "reads_per_s_bases_to_prefilter": nmap / (t_map + t_rest),
                "alignments_recovered": int(n_rec2), "good_reads": am.n_good(),
                "prefilter_best_is_truth": bool(tuple(int(x) for x in gts[int(np.argmax(sc_m))]) == tuple(Lo.true_genotype)),
                "truth_scores_as_the_best": bool(max(float(sc_m[i]) for i, g in enumerate(gts) if tuple(int(x) for x in g) == tuple(Lo.true_genotype)) >= float(sc_m.max()) - 1e-9 * abs(float(sc_m.max()))),
                "roofline": {"bound": "hbm", "kernel": "map_long_align_kernel", "achieved": per_aln * n_mapped / (ms_map * 1e-3) / 1e9 if ms_map else None,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": per_aln * n_mapped / (ms_map * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_map else None,
                             "traffic": long_route_traffic(n_mapped), "traffic_source": "profiles/r03_pmc_map_long_2048_ont_reads_x16.json (FETCH_SIZE raw + WRITE_SIZE per alignment of the align kernel, 10-kb reads)",
                             "algorithmic_bytes_per_alignment": per_aln,
                             "what": "read bases (2 bit) + allele bases under the read + CIGAR words out, per alignment; the kernel is bound by instruction issue "
                                     "(a row of the band per ~150 instructions of one wavefront), not by these bytes: DESIGN.md section 5"}}
            am.close(); del fq

    if args.ont_stream_sample > 0 and world == 1:
        progress("configs[2] from bases, streamed")
        from locityper_amd import legs
        ctx.trim()
        out["long_reads_stream"], (Ls, ps) = legs.ont_from_bases_stream(ctx, args.ont_stream_sample, A, progress=progress)
        if args.cpu_sample > 0:
            # the CPU beside it: the oracle has no restatement of the mapper in C (tests/pyref_map_long.py is Python: 16 alignments/s); what
            # it has for long reads is the reference's own route once a mapper has placed a read — AllAlignments::load with alignment recovery
            # onto the other alleles (locs.rs:1085-1185, transfer.rs:70-140) — timed on one core
            from tests import oracle_ffi as O
            ns = min(8, args.ont_stream_sample)
            ols = O.OracleLocus(Ls.seqs, Ls.seq_off, Ls.counts, Ls.cnt_off, Ls.k, Ls.bg, ps)
            Hos = O.HapAlns(A, transfer_fails=100, max_div=0.1)
            for q, r, w, _, _ in Ls.hap_alns(): Hos.add(q, r, w)
            Hos.sort()
            prim_s = Ls.reads(0, ns, primaries_only=True)
            tc = time.perf_counter()
            oas = ols.load_recover(prim_s, Hos)
            dtc = time.perf_counter() - tc
            out["long_reads_stream"]["cpu_baseline"] = {
                "value": ns / dtc, "unit": "reads/s", "cores": 1, "kind": "port",
                "sample": f"{ns} of those reads with the generator's primary record: oracle AllAlignments::load with alignment recovery onto the other "
                          f"{A - 1} alleles (the reference's route behind its mapper; the mapper itself has no C restatement)",
                "alignments_per_s": ns * A / dtc, "good_reads": oas.n_good}
            del ols, Hos, oas

    if first is not None:
        progress("CPU baseline")
        out["cpu_baseline"] = cpu_baseline(args, L, params, first, aa, gts, all_ixs, greedy, anneal, G, loc)
        out["vs_cpu_baseline"] = {k: reads_per_s / v["value"] for k, v in out["cpu_baseline"]["by_threads"].items()}
        cc = out["cpu_baseline"].pop("chains_check")
        out["chains_equal_oracle"] = None if cc is None else cc["chains_equal_oracle"]
        out["chains_check"] = cc
        from tests import oracle_ffi as O
        if out.get("recruitment") and args.recruit_sample > 0:
            # recruitment on the same core: the oracle's recruit_read_pair on a bounded sample of random pairs
            ot = O.OracleTargets(rprm.minimizer_k, rprm.minimizer_w, rprm.match_frac, rprm.match_length, rprm.thresh_kmer_count)
            ot.add_locus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k)
            ot.finalize()
            nsq = 20000
            acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
            rq_words = np.random.default_rng(11).integers(0, 1 << 32, size=nsq * 20, dtype=np.uint64).astype(np.uint32).reshape(nsq, 2, 10)
            codes = ((rq_words[..., None] >> (2 * np.arange(16, dtype=np.uint32))) & 3).reshape(nsq, 2, 160)[:, :, :150]
            sq = acgt[codes]
            tc = time.perf_counter()
            for i in range(nsq):
                ot.recruit(sq[i, 0].tobytes(), sq[i, 1].tobytes())
            out["cpu_baseline"]["recruitment_read_pairs_per_s"] = nsq / (time.perf_counter() - tc)
    if args.distinct_loci >= 2 and queue_mode and world == 1 and args.format == "counted":
        progress("queue of distinct loci, uploads inside the steps")
        out["distinct_loci_queue"] = distinct_loci_leg(args, ctx, loci, batches, stages, gts, ms_per_step, host_chunks)
    real_stdout.write(json.dumps(out) + "\n")
    real_stdout.flush()


if __name__ == "__main__":
    main()
