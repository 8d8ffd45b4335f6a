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

Prints ONE JSON line on rank 0. The legs beside the timed region live in bench_legs/ (its __init__ lists them).
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
from scripts.sources_sha import sources_sha16  # noqa: E402
from bench_legs import rooflines as RL  # noqa: E402
from bench_legs.common import HBM_PEAK_GBS, genotype_of, physical_cores, progress, survey_bytes_per_pair  # noqa: E402


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16,
                    help="loci of the timed queue (its last locus finishes alone: 0.33 s of drain shared by all)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=1_000_000, help="read pairs per locus (BASELINE: 1M)")
    ap.add_argument("--alleles", type=int, default=256)
    ap.add_argument("--chunk", type=int, default=32768, help="pairs per generated/uploaded chunk")
    ap.add_argument("--knob", action="append", default=[],
                    help="developer experiments: name=value for lcty_ctx_set_knob (repeatable)")
    ap.add_argument("--diag", action="store_true",
                    help="load the developer build of the library (make -C locityper_amd/csrc DIAG=1): trace and timing knobs exist there only")
    ap.add_argument("--cpu-sample", type=int, default=262144,
                    help="read pairs of the CPU baseline's load leg (0 = no CPU baseline); its run_filter and its solver chains "
                         "take the whole batch")
    ap.add_argument("--ont-sample", type=int, default=6144,
                    help="reads of the long-read recovery measurement (BASELINE.json configs[2] shape: 10 kb ONT reads x the locus' "
                         "alleles, the mapper reports the primaries, the other alleles are reached by alignment recovery); 0 = skip")
    ap.add_argument("--ont-whole-path-sample", type=int, default=8192,
                    help="10-kb ONT reads x the locus' alleles with their alignments GIVEN, through the whole path: score -> recover -> "
                         "score -> run_filter -> default solver scheme -> call (BASELINE.json configs[2] as the reference runs it); 0 = skip")
    ap.add_argument("--many-alleles-sample", type=int, default=65536,
                    help="read pairs of the extra measurement at 4 096 alleles (BASELINE.json configs[4], one GPU's shard): scoring and "
                         "the prefilter as f64 tile kernel and as integer Gram contraction on the matrix cores; 0 = skip")
    ap.add_argument("--format", choices=("counted", "records"), default="counted",
                    help="how the alignment table reaches the library: 16-byte counted alignments (lcty_reads_append_counted, SURVEY "
                         "8(d)'s alignment-table entry; the default) or BAM records with their CIGAR words (lcty_reads_append)")
    ap.add_argument("--cpu-reps", type=int, default=3, help="repetitions of the CPU baseline's greedy-chain figure (the median is reported)")
    ap.add_argument("--traffic", default="",
                    help="per-launch HBM bytes from the PMC passes (scripts/pmc_summary.py); default: the newest profiles/r*_pmc_traffic.json; "
                         "used when it matches the workload")
    ap.add_argument("--no-solve", action="store_true", help="leave the solver stages out of the step (score + prefilter only)")
    ap.add_argument("--shard-reads", action="store_true",
                    help="one locus over all ranks, whole path (BASELINE configs[4]): every rank scores and prefilters a contiguous shard "
                         "of the read pairs, the run_filter scores are SUM-all-reduced on the devices (RCCL), truncate_ixs runs "
                         "everywhere, every solver stage all-gathers the location-table rows of its alleles and deals its chains to the "
                         "ranks (lcty_solve_stage_read_sharded); strong scaling. Not the default: the driver's runs are one locus per rank")
    ap.add_argument("--shard-chains", action="store_true",
                    help="one locus over all ranks, whole path: every rank scores and prefilters all reads of the locus (replicated), the "
                         "(genotype, attempt) chains of both solver stages are dealt to the ranks and their likelihoods all-gathered on "
                         "the devices (RCCL; lcty_solve_stage_sharded, SURVEY 8e level 3); strong scaling. Not the default either")
    ap.add_argument("--distinct-loci", type=int, default=2,
                    help="extra measurement: a queue of loci that are NOT resident — this many distinct loci in page-locked host memory, "
                         "every position of the queue uploaded while the position before it is solved, three batch objects rotating "
                         "(lcty_solve_queue_fed); 0 = skip. The default, 2, takes the two loci of the main measurement")
    ap.add_argument("--distinct-steps", type=int, default=0,
                    help="positions of the timed queue of the --distinct-loci measurement (0: as many as --steps)")
    ap.add_argument("--loci-seeds", default="", help="developer measurement: seed offsets of the resident loci, comma-separated (default 0,1)")
    ap.add_argument("--distinct-no-upload", action="store_true",
                    help="developer measurement: the rotation of three batch objects through lcty_solve_queue_fed WITHOUT the uploads")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="allow more ranks than devices (launch-path checks on a one-GPU box; reported in the line)")
    ap.add_argument("--pipeline", type=int, default=0, help="ignored (round 1 option; the queue of loci is the default mode now)")
    ap.add_argument("--recruit-sample", type=int, default=8_000_000,
                    help="read pairs of the extra recruitment measurement (the step before the path, SURVEY 8f rank 1; 0 = skip)")
    ap.add_argument("--ont-map-sample", type=int, default=2048,
                    help="of the --ont-sample reads: mapped from their bases alone onto every allele (long route of candidate generation), "
                         "then scored and prefiltered (0 = skip)")
    ap.add_argument("--map-sample", type=int, default=32768,
                    help="read pairs mapped onto 8 basis alleles by the candidate-generation slice (0 = skip)")
    ap.add_argument("--ont-stream-sample", type=int, default=16384,
                    help="10-kb ONT reads of the configs[2] leg from bases alone, streamed (mapped onto all alleles on the device, scored, "
                         "prefiltered; 0 = skip)")
    ap.add_argument("--exact-sample", type=int, default=1,
                    help="extra measurement: the exact solver on BASELINE configs[0] (10 000 read pairs x 8 alleles, all 36 genotypes); 0 = skip")
    ap.add_argument("--exact-highs", type=int, default=0,
                    help="beside that leg, in the CPU block: HiGHS (scipy.optimize.milp) on the reference's programme for this many of the best "
                         "genotypes (half a minute to a minute and a half each on a busy host; tests/test_exact_highs.py holds the two "
                         "solvers against each other in any case)")
    ap.add_argument("--recovery-sample", type=int, default=262144,
                    help="read pairs of the extra alignment-recovery measurement (K6, outside the timed region; 0 = skip)")
    return ap.parse_args()


def set_up_ranks(args):
    """rank, world, the gloo group (barrier + max only), the context of this rank's device, knobs."""
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
        raise RuntimeError(f"{world} ranks but {ndev} HIP device(s) visible: one process per GPU (--oversubscribe puts several ranks on a "
                           "device to exercise the launch path on a small box; the line then says so)")
    ctx = api.Context(local_rank % ndev)
    if world > 1:          # this rank's share of the host's cores (spawn_ranks sets it; under torchrun: cores / ranks)
        ctx.set_knob("host_threads", max(1, min(16, int(os.environ.get("LCTY_BENCH_HOST_THREADS", physical_cores() // world)))))
    early_head = True      # lcty_solve_queue's default (knob queue_early_head)
    for kv in args.knob:
        name, _, val = kv.partition("=")
        ctx.set_knob(name, int(val))
        if name == "queue_early_head":
            early_head = int(val) != 0
    return rank, world, dist, ndev, ctx, early_head


def set_up_comm(args, ctx, rank, world, dist):
    """--shard-reads / --shard-chains: one RCCL communicator over the ranks (rank 0's id goes round through gloo)."""
    os.environ.pop("NCCL_DEBUG", None)         # RCCL logs to stdout, which carries the one JSON line
    os.environ["NCCL_DEBUG_FILE"] = os.devnull
    args.recovery_sample = args.recruit_sample = args.ont_sample = args.ont_stream_sample = args.ont_whole_path_sample = args.exact_sample = 0
    uid = api.comm_unique_id() if rank == 0 else bytes(api.COMM_ID_BYTES)
    if dist is not None:
        import torch
        t_uid = torch.tensor(list(uid), dtype=torch.uint8)
        dist.broadcast(t_uid, src=0)
        uid = bytes(t_uid.tolist())
    comm = api.Comm(ctx, world, rank, uid)
    return comm, comm.rccl_ranks()[0]


def load_loci(args, ctx, n_loci, rank, world, one_locus, first_pair, total_pairs):
    """Synthetic loci + reads (seed + locus index) -> HBM, chunk by chunk."""
    A = args.alleles
    loci, batches, host_chunks = [], [], []
    locus_setup_s = 0.0
    totals = {"recs": 0, "cigar": 0, "bases": 0}
    first = None
    params = None
    counted = args.format == "counted"
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
        aa = api.AllAlignments(loc, args.pairs, cap_bases, int(dens_r * args.pairs * head) + 4096,
                               0 if counted else int(dens_c * args.pairs * head) + 65536)
        # the chunks of the resident loci stay on the host when the queue of NON-resident loci is measured afterwards (it uploads them again)
        keep_host = args.distinct_loci >= 2 and not one_locus and not args.no_solve and world == 1 and counted
        host_chunks.append([c0] if keep_host else None)
        aa.append(c0, counted=counted)
        if j == 0:
            totals.update(recs=len(c0.recs), cigar=len(c0.cigar), bases=c0.n_bases)
            first = c0 if (rank == 0 and world == 1 and args.cpu_sample > 0) else None
        for ci in range(1, n_chunks):
            lo = ci * args.chunk
            ch = L.reads(first_pair + lo, min(args.chunk, args.pairs - lo))
            aa.append(ch, counted=counted)
            if j == 0:
                totals["recs"] += len(ch.recs)
                totals["cigar"] += len(ch.cigar)
                totals["bases"] += ch.n_bases
            if keep_host:
                host_chunks[j].append(ch)
            del ch
        loci.append((L, loc))
        batches.append(aa)
    return loci, batches, host_chunks, params, first, totals, locus_setup_s


def records_format_figure(args, ctx, loc, chunks, totals):
    """The scoring kernels on the RECORDS form of the same batch (BAM records + CIGAR words, lcty_reads_append: a8's CIGAR counting
    inside the kernel), alone on the device: the figure beside the counted form of the headline (verdict r05, weak 4)."""
    head = 1.03
    cap_bases = (int(totals["bases"] * head) + 1024) // 32 * 32 + 32
    ab = api.AllAlignments(loc, args.pairs, cap_bases, int(totals["recs"] * head) + 4096, int(totals["cigar"] * head) + 65536)
    for ch in chunks:
        ab.append(ch)
    ab.score()                                             # first use of the records kernels: code objects
    ctx.synchronize()
    ctx.timing_reset()
    ab.score()
    ctx.synchronize()
    n, ms = ctx.timing(api.K_SCORE)
    n_good = ab.n_good()
    ab.close()
    alg = survey_bytes_per_pair(args.alleles) * args.pairs
    rec_bytes = alg + 4.0 * totals["cigar"]               # the 16-byte table entry is the record; plus its CIGAR words
    return {"ms_alone": ms, "launches": int(n), "good_pairs": int(n_good),
            "frac_alone_survey_bytes": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms else None,
            "frac_alone_with_cigar_words": rec_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms else None,
            "cigar_words": totals["cigar"],
            "what": "lcty_score_reads on the same read pairs as 16-byte BAM records + CIGAR words (the kernel counts the operations: "
                    "Alignment::count_region_operations_fast, aln.rs:301-317), nothing else on the device; SURVEY 8(d)'s 12 331 B per pair, "
                    "and the same plus 4 B per CIGAR word"}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from bench_legs.launch import spawn_ranks
        spawn_ranks(args, __file__)     # never returns; nothing above this line loads the HIP library
    # stdout carries exactly one JSON line: whatever the libraries underneath print there (gloo announces its connections, RCCL its
    # version) goes to stderr, the line itself to the real stdout at the very end
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if args.diag:
        _lib.use_diag_build()
    _lib.lib()      # load the HIP library before anything else can bring another HIP runtime into scope
    rank, world, dist, ndev, ctx, early_head = set_up_ranks(args)

    comm = None
    rccl_ranks = None
    first_pair = 0
    total_pairs = args.pairs
    one_locus = args.shard_reads or args.shard_chains
    if one_locus:
        comm, rccl_ranks = set_up_comm(args, ctx, rank, world, dist)
    if args.shard_reads:
        per = (args.pairs + world - 1) // world
        first_pair = min(rank * per, args.pairs)
        args.pairs = min(first_pair + per, total_pairs) - first_pair          # this rank's shard

    # The default mode runs a QUEUE of loci through lcty_solve_queue (the loop of `locityper genotype` over its loci): the library
    # overlaps the last stage of a locus (annealing) with the scoring / prefilter / greedy stage of the next one, which needs two
    # loci resident; a step = one locus through the whole path, K steps = a queue of K loci alternating between the two.
    progress("generating the loci and their read pairs")
    t0 = time.time()
    n_loci = 1 if (one_locus or args.no_solve) else 2
    A = args.alleles
    loci, batches, host_chunks, params, first, totals, locus_setup_s = load_loci(args, ctx, n_loci, rank, world, one_locus, first_pair,
                                                                                total_pairs)
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

    def step(it=0, aa=aa):
        """One locus through the path, call by call (per-stage wall times; the modes that shard one locus over the ranks)."""
        t0s = time.perf_counter()
        aa.score()
        aa.prefilter_async()
        if args.shard_reads and aa is aa_main:
            comm.prefilter_allreduce(aa)                                      # read shards -> scores of the whole batch on every rank
        keep = aa.prefilter_truncate(params.filt_diff, 5000, params.threads)  # truncate_ixs, in_size of stage 1 (solve.rs:216-221)
        stage_s["score_prefilter"] += time.perf_counter() - t0s
        if args.no_solve:
            return keep, None
        # default scheme "-S greedy:i=5k,a=1 -S anneal:i=20,a=20" (solve.rs:211-230), then the final comparison
        n = len(gts)
        mean, var, att = np.full(n, np.nan), np.full(n, np.nan), np.zeros(n, dtype=np.uint32)
        ixs = keep
        # --shard-chains: the same call on every rank, the chains dealt to the ranks inside the library
        # --shard-reads: every rank holds its shard of the locus' reads; a stage exchanges the location-table rows of its alleles
        # (lcty_solve_stage_read_sharded: RCCL all-gathers), then deals its chains to the ranks like --shard-chains
        sharded = one_locus and aa is aa_main
        run_stage = (comm.solve_stage if args.shard_chains else comm.solve_stage_read_sharded) if sharded else api.solve_stage
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
        stage_s["greedy"] += tm - ts
        stage_s["anneal"] += te - tm
        return keep, res

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
    for k in stage_s:
        stage_s[k] = 0.0
    for k in solved:
        solved[k] = 0
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
    timers = RL.read_timers(ctx)
    calls_ok = None
    alone_ms = {}
    if queue_mode:
        calls = result
        called = genotype_of(gts, calls[-1].ixs[0])
        truth = loci[order[-1]][0].true_genotype
        calls_ok = all(genotype_of(gts, c.ixs[0]) == tuple(loci[j][0].true_genotype) for c, j in zip(calls, order))
        kept = int(calls[-1].kept_after_filter)
        quality = float(calls[-1].quality)
        greedy_chains_per_step = float(np.mean([int(c.kept_after_filter) for c in calls]))
        # per-stage wall times and iteration counts: one more locus call by call, outside the timed region
        if rank == 0:
            ctx.timing_reset()          # the queue's timers have been read: what follows are the kernels with nothing beside them
            step(7)
            ctx.synchronize()
            for name, (nk, msk) in RL.read_timers(ctx).items():
                if nk:
                    alone_ms[name] = msk / nk
    else:
        keep, res = result
        top = int(keep[0]) if res is None else int(res[0][0])
        called = genotype_of(gts, top)
        truth = L.true_genotype
        kept = int(len(keep))
        quality = None if res is None else float(res[2])
        greedy_chains_per_step = solved["greedy_chains"] / max(args.steps, 1)

    if rank != 0:
        return

    n_break = 1 if queue_mode else args.steps                  # steps behind stage_s / solved
    ms_per_step = 1e3 * elapsed / args.steps
    reads_per_s = (total_pairs if one_locus else world * args.pairs) * args.steps / elapsed
    n_score, ms_score = timers["score_reads_kernel"]
    n_pref, ms_pref = timers["prefilter_tile_kernel"]
    score_ms = ms_score / max(n_score, 1)
    pref_ms = ms_pref / max(n_pref, 1)
    layout_bytes = (totals["bases"] / 4 + totals["bases"] / 8 + 16 * totals["recs"]
                    + (0 if args.format == "counted" else 4 * totals["cigar"]) + 8 * A * args.pairs
                    + 8 * 4 * args.pairs)      # what the kernel's inputs/outputs occupy, excl. pair-alignment arena
    n_good = aa.n_good()
    per_step = {k: v / max(n_break, 1) for k, v in solved.items()}
    sha16 = sources_sha16(ROOT)
    # ---- rooflines of the kernels of a step; `roofline` is the one with the most kernel time (DESIGN.md §4 for the bytes) ----
    roofs = RL.kernel_rooflines(ROOT, timers, args.steps, args.pairs, A, G, n_good, per_step, queue_mode and early_head)
    # the dominant kernel of a step: the one with the most time on the MAIN stream, whose kernels run back to back and make up the
    # step; the annealing chains of the locus before run on the side stream next to them (overlapped, never on the critical path of
    # the queue) and are reported in roofline_all like everything else
    dominant = max((k for k in roofs if roofs[k]["bound"] == "hbm" and k != "anneal_loop_kernel"), key=lambda k: roofs[k]["ms_per_step"])
    if queue_mode:
        RL.attach_alone(roofs, alone_ms)
    traffic_path = args.traffic or RL.newest(ROOT, "r*_pmc_traffic.json")
    traffic_info = RL.attach_traffic(ROOT, roofs, traffic_path, args.pairs, A, sha16) if traffic_path else {}
    RL.attach_sq(ROOT, roofs, sha16)
    dom = roofs[dominant]
    workload = (f"{args.pairs} synthetic 150 bp PE read pairs x {A} alleles, 1 locus per step, k=25 "
                + ("(BASELINE.json configs[1])" if (total_pairs, A) == (1_000_000, 256) and not one_locus
                   else "(one GPU's share of BASELINE.json configs[4])" if A == 4096 else "(not a BASELINE.json configuration)"))
    if args.shard_reads:
        parallelism = (f"reads of one locus x{world}: RCCL all-reduce of the run_filter scores, all-gather of the location-table rows per "
                       "solver stage, chains dealt to the ranks")
    elif args.shard_chains:
        parallelism = f"solver chains of one locus x{world} (reads replicated), RCCL all-gather of the chain likelihoods"
    else:
        parallelism = f"loci x{world}"
    if queue_mode:
        step_what = ("one locus through lcty_solve_queue (score + run_filter + default solver scheme + final comparison); the queue "
                     "alternates between two resident loci; beside the greedy chains of a locus run the annealing stage of the locus "
                     "before (side stream) and"
                     + (" the scores, run_filter and location table of the locus after (fore stream)" if early_head else " nothing else"))
    else:
        step_what = "one locus, call by call"
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
        "config": {"workload": workload, "read_pairs": args.pairs, "alleles": A, "genotypes": G, "k": 25,
                   "records": totals["recs"], "cigar_words": totals["cigar"],
                   "alignment_table": ("16-byte counted alignments (lcty_reads_append_counted): the caller counts the CIGAR operations"
                                       if args.format == "counted" else "16-byte BAM records + CIGAR words (lcty_reads_append)"),
                   "step": step_what, "parallelism": parallelism},
        "reads_scored_per_s": (total_pairs if args.shard_reads else args.pairs) * n_break / max(stage_s["score_prefilter"], 1e-9),
        "genotypes_prefiltered_per_s": G * n_break / max(stage_s["score_prefilter"], 1e-9),
        "prefilter_genotypes_per_s_kernel": G / (pref_ms * 1e-3) if pref_ms else None,
        "kernel_ms_per_step": ({k: r["ms_per_step"] for k, r in roofs.items()}
                               | {"build_loc_table_kernel": timers["build_loc_table_kernel"][1] / args.steps}),
        "solver": None if args.no_solve else {
            "scheme": "greedy:i=5k,a=1 -> anneal:i=20,a=20 -> final comparison",
            "genotypes_solved_per_s": world * (greedy_chains_per_step + 20) * args.steps / elapsed,
            "chains_per_s": world * (greedy_chains_per_step + 400) * args.steps / elapsed,
            "per_step": per_step,
            "call_by_call_stage_ms": {k: 1e3 * v / max(n_break, 1) for k, v in stage_s.items()},
            "quality": quality, "all_calls_equal_truth": calls_ok},
        "roofline": {"bound": "hbm", "kernel": dominant, "achieved": dom["achieved"], "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": dom["frac"], "traffic": dom.get("traffic"),
                     "algorithmic_bytes_per_launch": dom["bytes"], "launch_ms": dom["launch_ms"], "what": dom["what"],
                     "traffic_fetch_raw": dom.get("traffic_fetch_raw"), "traffic_fetch_x2": dom.get("traffic_fetch_x2"),
                     "traffic_write": dom.get("traffic_write"), "traffic_rule": dom.get("traffic_rule"), "sq": dom.get("sq"),
                     **traffic_info},
        "roofline_all": roofs,
        "roofline_score_layout": {"layout_bytes_per_launch": layout_bytes,
                                  "achieved_layout_GBs": layout_bytes / (score_ms * 1e-3) / 1e9 if score_ms else None},
        "kernel_sources_sha16": sha16,
        "called_genotype": called, "true_genotype": truth, "kept_after_prefilter": kept,
        "setup_s": {"generate_and_upload": gen_s, "locus_create": locus_setup_s},
    }

    extra = world == 1
    if extra:
        progress("timed region done")
        ctx.trim()          # the solver workspaces of the timed steps (160 GB) make room for the extra measurements below
    rprm = None
    exact_state = None
    if extra and args.format == "counted" and host_chunks and host_chunks[0] and not args.no_solve:
        progress("scoring on the records form of the same batch")
        out["roofline_all"]["score_reads_kernel"]["records_format"] = records_format_figure(args, ctx, loc, host_chunks[0], totals)
        out["score_reads_kernel_records_ms_alone"] = out["roofline_all"]["score_reads_kernel"]["records_format"]["ms_alone"]
    if extra and args.recruit_sample > 0:
        progress("recruitment leg")
        from bench_legs.short_reads import recruitment_leg
        out["recruitment"], rprm = recruitment_leg(args, ctx, L)
    if extra and args.map_sample > 0:
        progress("candidate-generation leg")
        from bench_legs.short_reads import candidate_generation_leg
        out["candidate_generation"] = candidate_generation_leg(args, ctx, L, loc)
    if extra and args.recovery_sample > 0:
        progress("alignment-recovery leg")
        from bench_legs.short_reads import recovery_leg
        out["recovery"] = recovery_leg(args, ctx, L, loc)
    if extra and args.exact_sample > 0:
        progress("exact-solver leg")
        from bench_legs.short_reads import exact_solver_leg
        out["exact_solver"], exact_state = exact_solver_leg(args, ctx)
    if extra and args.many_alleles_sample > 0:
        progress("many-alleles leg")
        from bench_legs.short_reads import many_alleles_leg
        out["many_alleles"] = many_alleles_leg(args, ctx)
    with_cpu = first is not None
    oracle_build = None
    if with_cpu:
        # the oracle is compiled -march=native on THIS host before its first use (the chain checks of the long-read leg load it too)
        from bench_legs.cpu import native_oracle
        oracle_build = native_oracle(ROOT)
    if extra and (args.ont_sample > 0 or args.ont_whole_path_sample > 0 or args.ont_stream_sample > 0):
        from bench_legs import long_reads as LR
        out["long_reads"] = {}
        if args.ont_sample > 0:
            progress("long-read leg: alignment recovery")
            out["long_reads"], (Lo, po, loco) = LR.recovery_leg(args, ctx, gts, ROOT, sha16)
            if args.ont_map_sample > 0:
                progress("long-read leg: from bases alone ((f)2: the build's own mapper)")
                out["long_reads"]["from_bases"] = LR.from_bases_leg(args, ctx, gts, ROOT, Lo, loco)
            del Lo, po, loco
        if args.ont_whole_path_sample > 0:
            progress("long-read leg: the whole path on given alignments")
            ctx.trim()
            checker = None
            if with_cpu:
                from bench_legs.cpu import whole_path_chains_check as checker
            out["long_reads"]["whole_path"] = LR.ont_whole_path(ctx, args.ont_whole_path_sample, A, progress=progress, checker=checker)
            ctx.trim()
            out["long_reads"]["whole_path_counted"] = LR.ont_whole_path(ctx, args.ont_whole_path_sample, A, progress=progress,
                                                                        checker=checker, counted=True)
        if args.ont_stream_sample > 0:
            progress("configs[2] from bases, streamed ((f)2: the build's own mapper)")
            ctx.trim()
            out["long_reads_stream"], (Ls, ps) = LR.ont_from_bases_stream(ctx, args.ont_stream_sample, A, progress=progress)
            if with_cpu:
                from bench_legs.cpu import long_read_recovery_baseline
                out["long_reads_stream"]["cpu_baseline"] = long_read_recovery_baseline(Ls, ps, A)
            del Ls, ps
    if with_cpu:
        progress("CPU baseline")
        from bench_legs import cpu as CPU
        out["cpu_baseline"] = CPU.cpu_baseline(args, oracle_build, L, params, first, aa, gts, all_ixs, greedy, anneal, G, loc)
        cc = out["cpu_baseline"].pop("chains_check")
        out["chains_equal_oracle"] = None if cc is None else cc["chains_equal_oracle"]
        out["chains_check"] = cc
        if rprm is not None:
            out["cpu_baseline"]["recruitment_read_pairs_per_s"] = CPU.recruitment_baseline(L, rprm)
        if exact_state is not None and args.exact_highs > 0:
            progress("HiGHS on the reference's programme beside the exact-solver leg")
            try:
                out["cpu_baseline"]["exact_against_highs"] = CPU.exact_against_highs(exact_state, args.exact_highs)
            except Exception as e:                     # an extra beside an extra: the line stands without it
                out["cpu_baseline"]["exact_against_highs"] = {"error": str(e)}
            out["exact_solver"]["against_highs"] = "cpu_baseline.exact_against_highs (the CPU block: HiGHS on the reference's programme)"
    exact_state = None
    if args.distinct_loci >= 2 and queue_mode and extra and args.format == "counted":
        progress("queue of distinct loci, uploads inside the steps")
        from bench_legs.loci_queue import distinct_loci_leg
        out["distinct_loci_queue"] = distinct_loci_leg(args, ctx, loci, batches, stages, gts, ms_per_step, host_chunks)
    real_stdout.write(json.dumps(out) + "\n")
    real_stdout.flush()


if __name__ == "__main__":
    main()
