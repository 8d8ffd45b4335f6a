"""BASELINE.json configs[2] — 1 M 10-kb ONT reads x 256 alleles, the long-read DP path — in four measurements (harness code: ctypes
calls into liblocityper_hip.so, nothing computed here; the oracle is never touched from this module):

    recovery_leg            primaries only, every other allele through HapAlns::transfer_alignments (K6: transfer_kernel)
    from_bases_leg          the reads as sequenced mapped onto EVERY allele on the device (long route), resident batch
    ont_from_bases_stream   the same through a streaming batch, to the prefilter call
    ont_whole_path          THE PATH THE REFERENCE RUNS: the alignments GIVEN (as minimap2 supplies them, genotype.rs:990-1002), then
                            score -> recover -> score -> run_filter -> default solver scheme -> call (round 6)
"""
import json
import os
import time

import numpy as np

from locityper_amd import api, cdefs, synth
from .common import HBM_PEAK_GBS, genotype_of


def ont_from_bases_stream(ctx, n_reads, n_alleles=256, chunk=1024, read_len=10_000, seed_off=77, progress=None):
    """BASELINE.json configs[2] as named — 10-kb ONT reads x 256 alleles, the long-read DP path — from the bases alone to the prefilter call,
    through a streaming batch: every chunk of reads as sequenced is mapped onto EVERY allele on the device (long route: seeds -> one chain
    per (allele, strand) -> banded gap-affine alignment; the reference runs minimap2 -N min(25 000, 4 x alleles) here, genotype.rs:990-1002),
    its records go straight into the batch and are scored (AllAlignments::load), then dropped; run_filter over all genotypes at the end.
    Returns the leg's figures and (locus generator, resolved parameters)
    for the caller's CPU baseline (the oracle is test infrastructure: nothing in this package touches it)."""
    say = progress or (lambda *_: None)
    A = n_alleles
    L = synth.SynthLocus(A, n_reads, seed=synth.SEED + seed_off, technology=cdefs.TECH_NANOPORE, read_len=read_len)
    p = api.resolve_params(api.default_params(), L.bg)
    loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    mp = api.map_params(long_reads=True)
    t0 = time.perf_counter()
    api.build_map_index(loc, list(range(A)), k=mp.k)
    t_index = time.perf_counter() - t0
    t0 = time.perf_counter()
    chunks = [synth.sequencer_orientation(L.reads(lo, min(chunk, n_reads - lo), primaries_only=True)) for lo in range(0, n_reads, chunk)]
    say(f"  index over {A} alleles {t_index:.2f} s, {len(chunks)} chunks of reads generated in {time.perf_counter() - t0:.1f} s")
    cb = max(int(c.n_bases) for c in chunks); rb = max(int(c.mate_len.sum()) for c in chunks)
    read_bases = sum(int(c.mate_len.sum()) for c in chunks)

    def batch():
        return api.AllAlignments(loc, n_reads, (cb + 2048) // 32 * 32, chunk * A * 2 + 1024, rb // 3 * A + 4096, streaming_chunk_pairs=chunk)
    aw = batch()                                                # first use: code objects and the mapper's buffers (tens of GB, once per context)
    api.map_append(aw, chunks[0], mp); aw.score()
    aw.close()
    aa = batch()
    ctx.synchronize()
    ctx.timing_reset()
    t_map = t_score = 0.0
    t0 = time.perf_counter()
    for c in chunks:
        t1 = time.perf_counter()
        api.map_append(aa, c, mp)
        t2 = time.perf_counter()
        aa.score()
        t_map += t2 - t1; t_score += time.perf_counter() - t2
    ctx.synchronize()
    t_stream = time.perf_counter() - t0
    t0 = time.perf_counter()
    sc = aa.run_filter()
    t_filter = time.perf_counter() - t0
    n_launch, ms_map = ctx.timing(api.K_MAP)
    _, ms_score = ctx.timing(api.K_SCORE)
    gts = api.generate_genotypes(A, 2)
    best = tuple(int(x) for x in gts[int(np.argmax(sc))])
    truth = tuple(int(x) for x in L.true_genotype)
    truth_ix = next(i for i, g in enumerate(gts) if tuple(int(x) for x in g) == truth)
    mean_len = read_bases / max(n_reads, 1)
    cells = float(n_reads) * A * mean_len * (2 * mp.band + 1)
    total = t_stream + t_filter
    peak = 39.3e3 / 12.0                                        # G nodes/s: 1 024 SIMDs x 16 lanes x 2.4 GHz integer lane-operations at ~12 per node
    out = {
        "sample": f"{n_reads} synthetic {read_len}-base ONT reads as sequenced (no records) x {A} alleles (BASELINE.json configs[2]), "
                  f"streamed in chunks of {chunk}: mapped onto all {A} alleles on the device (seeds of {mp.k} every {mp.stride} bases, one "
                  f"chain per (allele, strand), gap-affine alignment in a band of +-{mp.band}), records straight into a streaming batch, "
                  f"scored chunk by chunk, prefiltered over all {len(gts)} genotypes",
        "reads": n_reads, "alleles": A, "chunk": chunk, "reads_per_s_bases_to_prefilter": n_reads / total, "seconds": total,
        "alignments": n_reads * A, "alignments_per_s": n_reads * A / total,
        "map_call_s": t_map, "score_call_s": t_score, "run_filter_s": t_filter, "index_build_s": t_index,
        "map_kernels_ms": ms_map, "score_kernels_ms": ms_score, "launches": int(n_launch),
        "alignments_per_s_map_kernels": n_reads * A / (ms_map * 1e-3) if ms_map else None,
        "aligned_bases_per_s_map_kernels": n_reads * A * mean_len / (ms_map * 1e-3) if ms_map else None,
        "good_reads": aa.n_good(), "best_genotype": list(best), "true_genotype": list(truth), "prefilter_best_is_truth": best == truth,
        "truth_scores_as_the_best": bool(sc[truth_ix] >= sc.max() - 1e-9 * abs(sc.max())),
        "roofline": {"bound": "valu_int", "kernel": "map_long_align_kernel", "unit": "G band nodes/s",
                     "achieved": cells / (ms_map * 1e-3) / 1e9 if ms_map else None, "peak": peak,
                     "frac": (cells / (ms_map * 1e-3) / 1e9) / peak if ms_map else None, "traffic": None,
                     "what": "nodes of the band of the gap-affine alignment (read bases x alleles x (2 * band + 1)) per second of the "
                             "mapper's kernels against the integer VALU rate (1 024 SIMDs x 16 lanes x 2.4 GHz) at ~12 lane-operations per "
                             "node; a band of +-16 keeps 33 of 64 lanes busy and a row costs ~40 vector + ~22 scalar instructions "
                             "(profiles/r03_pmc_map_long_2048_ont_reads_x16.json); HBM traffic is ~120 KB per alignment, small beside it"},
    }
    aa.close()
    return out, (L, p)


def recovery_leg(args, ctx, gts, root, sha16):
    """configs[2] shape, the long-read DP path of the reference: 10-kb single-end ONT reads, primaries only, every other allele reached by
    HapAlns::transfer_alignments (two-CIGAR walk + gap-affine aligner on the stretches between anchors). Returns the leg and what the
    from-bases leg re-uses (generator, locus, the primaries)."""
    A, nont = args.alleles, args.ont_sample
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
    ao.recover()                                         # the first call allocates the lane scratch of the context (tens of GB)
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
    leg = {"sample": f"{nont} synthetic 10-kb ONT reads x {A} alleles (BASELINE.json configs[2] shape), primary records only, "
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
    leg["prefilter_best_is_truth"] = bool(genotype_of(gts, np.argmax(sc_o)) == tuple(Lo.true_genotype))
    # per transfer the kernel has to look at the read's CIGAR (4 B per item), the part of the haplotype-to-haplotype CIGAR under the
    # read (8 B per item), the target's bases under the read, and write the transferred CIGAR (4 B per item): its algorithmic bytes
    cig_items = float(sum(len(c.cigar) for c in prim)) / max(sum(c.n_pairs for c in prim), 1)
    hap_items = float(np.mean([len(h[2]) for h in Ho[:512]])) * 10_000.0 / float(np.mean(np.diff(Lo.seq_off)))
    per_transfer = 4.0 * cig_items + 8.0 * hap_items + 10_000.0 + 4.0 * cig_items
    gbs = per_transfer * n_new / (ms_tr * 1e-3) / 1e9 if ms_tr else None
    leg["roofline"] = {"bound": "hbm", "kernel": "transfer_kernel", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": gbs / HBM_PEAK_GBS if gbs else None, "traffic": None, "algorithmic_bytes_per_transfer": per_transfer,
                       "what": "read CIGAR + haplotype CIGAR under the read + target bases under the read + transferred CIGAR, per transfer"}
    try:
        # HBM bytes per transfer from the committed counter passes over the same kernel (scripts/run_transfer_traffic.sh)
        from .rooflines import newest
        path = newest(root, "r*_pmc_transfer_kernel.json")
        tt = json.load(open(path))
        bpt = tt["bytes_per_transfer"]
        leg["roofline"].update({"traffic": (bpt["fetch_raw"] + bpt["write"]) * n_new, "traffic_bytes_per_transfer": bpt,
                                "traffic_rule": "FETCH_SIZE + WRITE_SIZE (narrow gathers: raw)",
                                "traffic_source": os.path.relpath(path, root), "traffic_is_current": tt.get("sources_sha16") == sha16})
    except (OSError, KeyError, ValueError, TypeError):
        pass
    ao.close()
    return leg, (Lo, po, loco)


def long_route_traffic(root, n_alignments):
    """HBM bytes of the long route's align kernel for that many alignments of 10-kb reads, from the committed counter passes."""
    try:
        with open(os.path.join(root, "profiles", "r03_pmc_map_long_2048_ont_reads_x16.json")) as f:
            t = json.load(f)["traffic"]["align_kernel_bytes_per_alignment"]
        return (t["fetch_raw"] + t["write"]) * n_alignments
    except (OSError, KeyError, ValueError):
        return None


def from_bases_leg(args, ctx, gts, root, Lo, loco):
    """The same shape from bases alone (second slice; lcty_map_long.hip): the reads as the sequencer gave them are
    mapped onto EVERY allele (seed - chain - gap-affine alignment along the chain; the reference runs minimap2 here, genotype.rs:990-1002,
    --basis none), the records go straight into a batch: per-read alignment against every allele on the device, no external mapper."""
    A = args.alleles
    ctx.trim()                                       # the solver workspaces of the timed region (150 GB) are not this leg's business
    nmap = min(args.ont_map_sample, args.ont_sample)
    chunk_m = 1024                                   # read ends per mapping call: one wavefront each in the chain kernel
    fq = [synth.sequencer_orientation(Lo.reads(lo, min(chunk_m, nmap - lo), primaries_only=True)) for lo in range(0, nmap, chunk_m)]
    mpl = api.map_params(long_reads=True)
    basis_l = list(range(A))
    tm0 = time.perf_counter()
    api.build_map_index(loco, basis_l, k=mpl.k)
    t_index = time.perf_counter() - tm0
    tot_bases = sum(int(c.n_bases) for c in fq)
    read_bases = sum(int(c.mate_len.sum()) for c in fq)
    cap_cig = int(read_bases // 3) * len(basis_l) + 4096
    am = api.AllAlignments(loco, nmap, (tot_bases + 2048) // 32 * 32, nmap * len(basis_l) * 2 + 1024, cap_cig)
    api.map_append(am, fq[0], mpl)                   # warm-up: code objects, and the buffers of the mapping, which stay with the context
    am.reset(loco)                                   # (tens of GB: the first allocation takes seconds); then the batch empty again
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
    mean_len = read_bases / max(nmap, 1)
    cells_m = float(n_mapped) * mean_len * (2 * mpl.band + 1)
    per_aln = mean_len * (0.25 + 1.0 + 1.0)          # packed read bases + allele bases under the read + ~a CIGAR word per 4 bases
    truth = tuple(Lo.true_genotype)
    truth_score = max(float(sc_m[i]) for i, g in enumerate(gts) if tuple(int(x) for x in g) == truth)
    gbs = per_aln * n_mapped / (ms_map * 1e-3) / 1e9 if ms_map else None
    leg = {
        "sample": f"the first {nmap} of those reads as sequenced (no records) onto all {len(basis_l)} alleles: seeds of {mpl.k} every "
                  f"{mpl.stride} bases, one chain per (allele, strand), gap-affine alignment along the chain in a band of +-{mpl.band}; records "
                  f"straight into a batch (lcty_reads_map_append, chunks of {chunk_m}), then scoring (+ recovery of the few alignments the "
                  "mapper left out) + prefilter",
        "alignments": n_mapped, "map_kernels_ms": ms_map, "launches": int(n_launch), "map_call_s": t_map, "index_build_s": t_index,
        "alignments_per_s_kernel": n_mapped / (ms_map * 1e-3) if ms_map else None, "reads_per_s_call": nmap / t_map,
        "aligned_bases_per_s_kernel": n_mapped * mean_len / (ms_map * 1e-3) if ms_map else None,
        "band_cells": cells_m, "gcups": cells_m / (ms_map * 1e-3) / 1e9 if ms_map else None,
        "score_recover_rescore_s": t_rest, "score_s": t_score1, "recover_s": t_recover, "run_filter_s": t_filter,
        "reads_per_s_bases_to_prefilter": nmap / (t_map + t_rest),
        "alignments_recovered": int(n_rec2), "good_reads": am.n_good(),
        "prefilter_best_is_truth": bool(genotype_of(gts, np.argmax(sc_m)) == truth),
        "truth_scores_as_the_best": bool(truth_score >= float(sc_m.max()) - 1e-9 * abs(float(sc_m.max()))),
        "roofline": {"bound": "hbm", "kernel": "map_long_align_kernel", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": gbs / HBM_PEAK_GBS if gbs else None, "traffic": long_route_traffic(root, n_mapped),
                     "traffic_source": "profiles/r03_pmc_map_long_2048_ont_reads_x16.json (FETCH_SIZE raw + WRITE_SIZE per alignment of "
                                       "the align kernel, 10-kb reads)",
                     "algorithmic_bytes_per_alignment": per_aln,
                     "what": "read bases (2 bit) + allele bases under the read + CIGAR words out, per alignment; the kernel is bound by "
                             "instruction issue (a row of the band per ~150 instructions of one wavefront), not by these bytes: DESIGN.md §5e"}}
    am.close()
    return leg


def ont_whole_path(ctx, n_reads, n_alleles=256, chunk=4096, read_len=10_000, seed_off=77, master_seed=4242, progress=None, checker=None,
                   counted=False):
    """BASELINE.json configs[2] on the path the reference runs (verdict r05, item 3). minimap2 hands `locityper genotype` an alignment of
    every read to every allele (`-N min(25 000, 4 x alleles)`, genotype.rs:990-1002); from there the reference goes
        AllAlignments::load (locs.rs:1085-1185; single-end grouping 873-911, unmapped_penalty 1e-100 for long reads, model/mod.rs:55-60)
        -> recover_and_group_alignments with the haplotype alignments (1237-1288)
        -> run_filter + truncate_ixs -> the default scheme (greedy 5 000 x 1, annealing 20 x 20) -> produce_result.
    Two forms of the same reads and alignments:
      records (default)  the generator's records — one alignment with its full CIGAR per read and allele: ~2.7 KB each, 700 KB per read,
                         700 GB for the configuration: it cannot be resident — are streamed through a streaming batch in chunks from
                         page-locked memory: append (host to device), lcty_score_reads, lcty_recover_alignments (which finds every
                         allele reached and carries over only the decoy alignments of 5 % of the reads), lcty_score_reads again;
                         then lcty_solve. Every call is on the clock, the uploads included; the generator is not.
      counted            the caller has counted the operations of every CIGAR (Alignment::count_region_operations_fast, aln.rs:301-317,
                         which read_next_alns does on the host anyway): 16 bytes per alignment, 4 GB for the configuration — the
                         batch is RESIDENT before the clock starts, as in the main measurement; score -> lcty_solve on the clock.
                         (Alignment recovery needs the CIGARs: not in this form.)
    `checker(L, p, loc, aa, gts, scores, greedy, anneal)`: the caller's comparison of eight chains with the oracle (bench_legs.cpu)."""
    say = progress or (lambda *_: None)
    A = n_alleles
    L = synth.SynthLocus(A, n_reads, seed=synth.SEED + seed_off, technology=cdefs.TECH_NANOPORE, read_len=read_len)
    p = api.resolve_params(api.default_params(), L.bg)
    loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    t_hap = 0.0
    H = None
    if not counted:
        t0 = time.perf_counter()
        H = L.hap_alns()
        loc.set_hap_alns(H, transfer_fails=100, max_div=0.1)
        t_hap = time.perf_counter() - t0
    chunk = min(chunk, n_reads)
    c0 = L.reads(0, chunk)
    f = 1.25 * (n_reads if counted else chunk) / c0.n_pairs
    cap_bases = (int(c0.n_bases * f) + 2048) // 32 * 32
    if counted:
        aa = api.AllAlignments(loc, n_reads, cap_bases, int(len(c0.recs) * f) + 1024, 0)
    else:
        aa = api.AllAlignments(loc, n_reads, cap_bases, int(len(c0.recs) * f) + 1024, int(len(c0.cigar) * f) + 65536,
                               streaming_chunk_pairs=chunk)
    ctx.synchronize()
    ctx.timing_reset()
    t_gen = t_app = t_score = t_rec = 0.0
    n_new = n_recs = n_cigar = 0
    up_bytes = 0
    for lo in range(0, n_reads, chunk):
        tg = time.perf_counter()
        c = c0 if lo == 0 else L.reads(lo, min(chunk, n_reads - lo))
        n_recs += len(c.recs)
        n_cigar += len(c.cigar)
        if counted:
            aa.append(c, counted=True)                      # resident before the clock starts
            t_gen += time.perf_counter() - tg
            del c
            continue
        pc = cdefs.ReadsChunk(c.mate_len, c.mate_off, ctx.pinned_like(c.bases2), c.nmask, c.aln_off, ctx.pinned_like(c.recs),
                              c.cigar_off, ctx.pinned_like(c.cigar))
        up_bytes += c.recs.nbytes + c.cigar.nbytes + c.bases2.nbytes + c.nmask.nbytes
        del c
        t1 = time.perf_counter()
        aa.append(pc)
        ctx.synchronize()
        t2 = time.perf_counter()
        aa.score()
        ctx.synchronize()
        t3 = time.perf_counter()
        n_new += aa.recover()                               # with its second scoring pass when anything was transferred
        ctx.synchronize()
        t4 = time.perf_counter()
        t_gen += t1 - tg
        t_app += t2 - t1
        t_score += t3 - t2
        t_rec += t4 - t3
        if lo:
            say(f"  whole path: {lo + pc.n_pairs} of {n_reads} reads scored")
        del pc
    c0 = None
    if counted:
        ctx.synchronize()
        t0 = time.perf_counter()
        aa.score()
        ctx.synchronize()
        t_score = time.perf_counter() - t0
    stages = api.default_stages()
    # the first lcty_solve of a context allocates the chains' workspace (150 GB at this size: seconds, once per context — the main
    # measurement's warm-up steps pay it there); the second call, on the workspace the library keeps, is the one on the clock
    t0 = time.perf_counter()
    api.solve_locus(aa, stages, master_seed)
    ctx.synchronize()
    t_solve_first = time.perf_counter() - t0
    before = {kid: ctx.timing(kid)[1] for kid in (api.K_PREFILTER, api.K_SOLVE_TABLE, api.K_SOLVE_INIT, api.K_SOLVE_INIT_ANNEAL, api.K_SOLVE,
                                                  api.K_ANNEAL)}
    t0 = time.perf_counter()
    call, mean, var, att = api.solve_locus(aa, stages, master_seed)
    ctx.synchronize()
    t_solve = time.perf_counter() - t0
    kern = {name: ctx.timing(kid)[1] - before.get(kid, 0.0) for name, kid in (
        ("score_reads_kernel", api.K_SCORE), ("transfer_kernel", api.K_TRANSFER), ("prefilter_tile_kernel", api.K_PREFILTER),
        ("build_loc_table_kernel", api.K_SOLVE_TABLE), ("solve_init_kernel", api.K_SOLVE_INIT),
        ("solve_init_kernel_annealing_stage", api.K_SOLVE_INIT_ANNEAL), ("greedy_loop_kernel", api.K_SOLVE),
        ("anneal_loop_kernel", api.K_ANNEAL))}
    gts = api.generate_genotypes(A, 2)
    truth = tuple(int(x) for x in L.true_genotype)
    called = genotype_of(gts, call.ixs[0])
    n_good = aa.n_good()
    total = t_app + t_score + t_rec + t_solve
    chains, iters, _ = api.solve_stats(aa)                  # of the last stage (annealing)
    kept = int(call.kept_after_filter)
    # algorithmic bytes per kernel. Scoring: per read its packed bases and the matrix row; per alignment the 16-byte table entry
    #  and — in the records form, where the kernel counts the operations itself — its CIGAR words
    score_bytes = n_reads * (read_len / 4.0 + 8.0 * A) + 16.0 * n_recs + (0.0 if counted else 4.0 * n_cigar)
    # the greedy stage: 5 000 chains (or as many as the prefilter kept) x 100 000 iterations x 10 candidates x one 32-byte record
    greedy_bytes = 32.0 * 10 * 100_000 * min(kept, 5000)
    roofs = {"score_reads_kernel": score_bytes, "greedy_loop_kernel": greedy_bytes, "anneal_loop_kernel": 32.0 * float(iters)}
    what = {"score_reads_kernel": "packed bases + the 16 B table entry" + ("" if counted else " + CIGAR words") + " of every alignment "
                                  "+ the matrix row, per read",
            "greedy_loop_kernel": "one 32 B record per candidate read, 10 candidates per iteration, 100 000 iterations per chain (an upper "
                                  "bound: chains that reach their plateau stop earlier)",
            "anneal_loop_kernel": "one 32 B record per evaluated move (latency-bound serial chains: 400 chains on 800 wavefronts)"}
    if H is not None:
        # a transfer: the read's CIGAR (4 B per item) + the haplotype CIGAR under the read (8 B per item) + the target's bases under the
        # read + the transferred CIGAR out (4 B per item) — the per-transfer figure of the recovery leg
        cig_items = n_cigar / max(n_recs, 1)
        hap_items = float(np.mean([len(h[2]) for h in H[:512]])) * read_len / float(np.mean(np.diff(L.seq_off)))
        roofs["transfer_kernel"] = (8.0 * cig_items + 8.0 * hap_items + read_len) * n_new
        what["transfer_kernel"] = ("read CIGAR + haplotype CIGAR under the read + target bases under the read + transferred CIGAR, per "
                                   "transfer (the decoy alignments of 5 % of the reads are carried to the other alleles)")
    dominant = max(kern, key=lambda k: kern[k])
    roof = {"kernel": dominant, "kernel_ms": kern[dominant], "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None}
    if dominant in roofs and kern[dominant] > 0:
        roof["algorithmic_bytes"] = roofs[dominant]
        roof["achieved"] = roofs[dominant] / (kern[dominant] * 1e-3) / 1e9
        roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
        roof["what"] = what[dominant]
    if counted:
        sample = (f"{n_reads} synthetic {read_len}-base ONT reads x {A} alleles (BASELINE.json configs[2] shape), one GIVEN alignment per read "
                  f"and allele as a 16-byte counted entry ({n_recs} entries; the caller counted the {n_cigar} CIGAR operations), the batch "
                  f"resident: lcty_score_reads, then lcty_solve: run_filter over {len(gts)} genotypes, truncate_ixs, greedy 5 000 x 1, "
                  "annealing 20 x 20, final comparison, unexplained reads")
    else:
        sample = (f"{n_reads} synthetic {read_len}-base ONT reads x {A} alleles (BASELINE.json configs[2] shape), one GIVEN alignment with "
                  f"its CIGAR per read and allele ({n_recs} records, {n_cigar} CIGAR words = {up_bytes / 1e9:.1f} GB uploaded from page-locked "
                  f"memory), streamed in chunks of {chunk}: append -> score -> recover (+ score) per chunk, then lcty_solve: run_filter over "
                  f"{len(gts)} genotypes, truncate_ixs, greedy 5 000 x 1, annealing 20 x 20, final comparison, unexplained reads")
    out = {
        "sample": sample, "alignment_table": "16-byte counted alignments, resident" if counted else "records + CIGAR words, streamed",
        "reads": n_reads, "alleles": A, "chunk": chunk, "good_reads": int(n_good),
        "reads_per_s": n_reads / total, "seconds": total,
        "reads_per_s_without_the_uploads": n_reads / (total - t_app),
        "append_s": t_app, "upload_GBs": up_bytes / 1e9 / t_app if t_app else None, "score_call_s": t_score,
        "recover_call_s": t_rec, "alignments_transferred": int(n_new), "solve_call_s": t_solve,
        "solve_first_call_s_with_the_workspace_allocation": t_solve_first,
        "kernel_ms": kern, "kept_after_filter": kept,
        "called_genotype": list(called), "true_genotype": list(truth), "all_calls_equal_truth": called == truth,
        "quality": float(call.quality), "unexpl_reads": int(call.unexpl_reads),
        "annealing_chains": int(chains), "annealing_moves": int(iters),
        "hap_alns_s": t_hap, "generator_s_not_timed": t_gen,
        "roofline": roof,
    }
    if checker is not None:
        scores = aa.run_filter()
        out["chains_check"] = checker(L, p, loc, aa, gts, scores, api.default_solver(cdefs.SOLVER_GREEDY), api.default_solver(cdefs.SOLVER_ANNEAL))
        out["chains_equal_oracle"] = out["chains_check"]["chains_equal_oracle"]
    aa.close()
    return out
