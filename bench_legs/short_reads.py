"""Extra measurements around the path on short reads (outside the timed region): read recruitment, candidate
generation on a basis (rank 2, first slice), alignment recovery (K6), the 4 096-allele shard of BASELINE.json configs[4]."""
import time

import numpy as np

from locityper_amd import api, synth
from locityper_amd.cdefs import ALN_REC_DTYPE, ReadsChunk


def recruitment_leg(args, ctx, L):
    """Minimizer read recruitment (Targets::recruit_read_pair, seq/recruit.rs:883-929), the step before the path: random 150 + 150-base
    pairs (whole-genome input is almost entirely foreign to a locus) against this locus' alleles, and among them 0.2 % pairs drawn from
    the locus itself — AS THE SEQUENCER GAVE THEM: recruitment reads FASTQ records (seq/fastx.rs:141-150), in which the mates of a
    pair face each other; the generator's chunks hold SEQ as a BAM does (reverse-complemented where the record is on the reverse
    strand), and recruit_read_pair's `better_pair_fraction` (recruit.rs:923-929) rejects a pair whose mates look the same way. The leg
    fails unless >= 85 % of the locus' own pairs and <= 0.01 % of the random ones are recruited."""
    A = args.alleles
    nrq = args.recruit_sample
    rprm = api.recruit_params()
    tq0 = time.perf_counter()
    T = api.Targets(ctx, rprm)
    T.add_locus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k)
    n_minim = T.finalize()
    t_targets = time.perf_counter() - tq0
    rngq = np.random.default_rng(11)
    words = rngq.integers(0, 1 << 32, size=nrq * 20, dtype=np.uint64).astype(np.uint32)
    n_own = max(1, nrq // 500)
    own = synth.sequencer_orientation(L.reads(0, min(n_own, args.pairs), primaries_only=True))
    full = np.flatnonzero((own.mate_len.reshape(-1, 2) == 150).all(axis=1))          # pairs of two whole 150-base mates
    n_own = len(full)
    own_at = rngq.choice(nrq, size=n_own, replace=False)
    mo = own.mate_off.astype(np.int64) // 16
    for t, pair in enumerate(full):
        for e in range(2):
            at = (2 * int(own_at[t]) + e) * 10
            words[at:at + 10] = own.bases2[mo[2 * pair + e]:mo[2 * pair + e] + 10]
    rq = ReadsChunk(np.full(2 * nrq, 150, dtype=np.uint32), np.arange(2 * nrq + 1, dtype=np.uint64) * 160, words,
                    np.zeros(nrq * 10, dtype=np.uint32), np.zeros(nrq + 1, dtype=np.uint64), np.zeros(0, dtype=ALN_REC_DTYPE),
                    np.zeros(nrq + 1, dtype=np.uint64), np.zeros(0, dtype=np.uint32))
    ctx.timing_reset()
    tq0 = time.perf_counter()
    cntq, _ = T.recruit(rq, paired=True)
    t_call = time.perf_counter() - tq0
    _, ms_q = ctx.timing(api.K_RECRUIT)
    hit = cntq != 0
    own_hit = int(np.count_nonzero(hit[own_at]))
    foreign_hit = int(np.count_nonzero(hit)) - own_hit
    leg = {"sample": f"{nrq} 150 + 150-base read pairs in FASTQ orientation, {n_own} of them drawn from the locus and the rest random, "
                     f"against its {A} alleles ({n_minim} minimizers)",
           "kernel_ms": ms_q, "read_pairs_per_s_kernel": nrq / (ms_q * 1e-3) if ms_q else None,
           "read_pairs_per_s_call": nrq / t_call, "targets_build_s": t_targets, "recruited": int(np.count_nonzero(hit)),
           "locus_derived_pairs": int(n_own), "locus_derived_recruited": own_hit,
           "locus_derived_recruited_frac": own_hit / max(n_own, 1),
           "random_pairs_recruited": foreign_hit, "random_pairs_recruited_frac": foreign_hit / max(nrq - n_own, 1)}
    T.close()
    if n_own >= 20 and own_hit < 0.85 * n_own:
        raise RuntimeError(f"recruitment leg: {own_hit} of the locus' own {n_own} read pairs recruited (at least 85 % expected)")
    if foreign_hit > 1e-4 * (nrq - n_own) + 1:
        raise RuntimeError(f"recruitment leg: {foreign_hit} of {nrq - n_own} random read pairs recruited (at most 0.01 % expected)")
    return leg, rprm


def candidate_generation_leg(args, ctx, L, loc):
    """Candidate generation on the basis alleles (first slice; lcty_map.hip), the step the reference leaves to an
    external mapper: the first read pairs of the locus, their bases only, onto 8 basis alleles."""
    A = args.alleles
    nmp = min(args.map_sample, args.pairs)
    src = L.reads(0, nmp)
    zoff = np.zeros(nmp + 1, dtype=np.uint64)
    bare = ReadsChunk(src.mate_len, src.mate_off, src.bases2, src.nmask, zoff, np.zeros(0, dtype=ALN_REC_DTYPE), zoff,
                      np.zeros(0, dtype=np.uint32))
    mp = api.map_params()
    basis = list(range(0, A, max(1, A // 8)))[:8]
    tm0 = time.perf_counter()
    api.build_map_index(loc, basis, k=mp.k)
    t_index = time.perf_counter() - tm0
    tm0 = time.perf_counter()
    mapped = api.map_reads(loc, bare, mp)              # to the host: sizes, then records (the chunk is mapped twice); warm-up
    t_host = time.perf_counter() - tm0
    # the path of a run: the records straight into a batch of the locus (device to device), one mapping
    am = api.AllAlignments(loc, nmp, (int(src.n_bases) + 2048) // 32 * 32, len(mapped.recs) + 1024, len(mapped.cigar) + 1024)
    ctx.timing_reset()
    tm0 = time.perf_counter()
    api.map_append(am, bare, mp)
    t_map = time.perf_counter() - tm0
    _, ms_map = ctx.timing(api.K_MAP)
    am.close()
    return {"sample": f"first {nmp} read pairs (bases only) onto {len(basis)} basis alleles, seeds of {mp.k} every {mp.stride} bases, ungapped "
                      "extension then a band alignment with gaps for clipped candidates; records straight into a batch (lcty_reads_map_append)",
            "records": int(len(mapped.recs)), "kernel_ms": ms_map,
            "read_ends_per_s_kernel": 2 * nmp / (ms_map * 1e-3) if ms_map else None,
            "read_ends_per_s_call": 2 * nmp / t_map, "to_host_two_mappings_s": t_host, "index_build_s": t_index}


def recovery_leg(args, ctx, L, loc):
    """Alignment recovery (K6), not part of the step: the mapper reports only the primary alignment of each read end, the other alleles
    are reached through the haplotype-to-haplotype alignments (transfer.rs:70-140)."""
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
    n_tr, ms_tr = ctx.timing(api.K_TRANSFER)
    cells = ab.recover_dp_cells()
    leg = {"aligner_cells": int(cells), "gcups": cells / (ms_tr * 1e-3) / 1e9 if ms_tr else None,
           "sample": f"first {nrec} read pairs, primary records only ({sum(len(c.recs) for c in prim)} records), {len(H)} haplotype "
                     "alignments, transfer_fails 100",
           "alignments_transferred": int(n_new), "transfer_kernel_ms": ms_tr, "launches": int(n_tr),
           "transfers_per_s": n_new / (ms_tr * 1e-3) if ms_tr else None,
           "recover_and_rescore_s": t_rec, "set_hap_alns_s": t_set, "good_pairs_after": ab.n_good(), "level_pairs": ab.recover_stats()}
    ab.close()
    return leg


def many_alleles_leg(args, ctx):
    """configs[4] shard shape: a locus of 4 096 alleles, the read pairs one of eight GPUs would hold. The prefilter is the dominant
    kernel there; from 512 alleles on it runs as an integer Gram contraction on the matrix cores (lcty_gram.hip)."""
    nma, Ama = args.many_alleles_sample, 4096
    Lm = synth.SynthLocus(Ama, nma, seed=synth.SEED + 5, base_len=3000)
    pm = api.resolve_params(api.default_params(), Lm.bg)
    locm = api.Locus(ctx, Lm.seqs, Lm.seq_off, Lm.counts, Lm.cnt_off, Lm.k, Lm.bg, pm)
    am = None
    for lo in range(0, nma, 4096):
        chm = Lm.reads(lo, min(4096, nma - lo))
        if am is None:
            fm = 1.1 * nma / chm.n_pairs
            am = api.AllAlignments(locm, nma, (int(chm.n_bases * fm) + 2048) // 32 * 32, int(len(chm.recs) * fm) + 4096,
                                   int(len(chm.cigar) * fm) + 65536)
        am.append(chm)
    ctx.timing_reset()
    am.score()
    ctx.synchronize()
    _, ms_score_m = ctx.timing(api.K_SCORE)
    leg = {"workload": f"{nma} read pairs x {Ama} alleles ({Ama * (Ama + 1) // 2} genotypes): BASELINE configs[4] gives each of eight "
                       "GPUs 625 000 read pairs of such a locus",
           "score_reads_kernel_ms": ms_score_m}
    scores_m = {}
    for name, knob in (("f64_tile_kernel", 0), ("integer_gram_on_mfma", 1)):
        ctx.set_knob("prefilter_gram", knob)
        am.prefilter_async()
        ctx.synchronize()
        ctx.timing_reset()
        am.prefilter_async()
        ctx.synchronize()
        leg[name + "_ms"] = ctx.timing(api.K_PREFILTER)[1]
        scores_m[name] = am.prefilter_scores()
    ctx.set_knob("prefilter_gram", -1)
    ref = scores_m["f64_tile_kernel"]
    leg["max_relative_difference"] = float(np.abs(scores_m["integer_gram_on_mfma"] - ref).max() / np.abs(ref).max())
    best_m = api.generate_genotypes(Ama, 2)[int(np.argmax(scores_m["integer_gram_on_mfma"]))]
    leg["best_genotype"] = [int(x) for x in best_m]
    leg["true_genotype"] = list(Lm.true_genotype)
    am.close()
    return leg


def exact_solver_leg(args, ctx):
    """The exact solver (the place of HighsSolver, solvers/highs.rs:38-134) on BASELINE.json configs[0] — 10 000 read pairs x 8
    alleles: every one of the locus' 36 genotypes, one attempt each, through lcty_solve_stage (models built on the device, solved by the
    pool of host threads). Returns the leg and what bench_legs/cpu.py::exact_against_highs needs to hold HiGHS beside it."""
    from locityper_amd import cdefs
    L = synth.SynthLocus(8, 10_000, seed=synth.SEED + 3)
    p = api.resolve_params(api.default_params(), L.bg)
    loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    aa = api.AllAlignments.load(loc, L.reads(0, 10_000))
    gts = api.generate_genotypes(8, 2)
    order = np.argsort(-aa.run_filter(), kind="stable")
    every = np.ascontiguousarray(gts[order])
    seeds = api.chain_seeds(77, len(every))
    ex = api.default_solver(cdefs.SOLVER_EXACT)
    api.solve_stage(aa, every[:1], ex, 1, seeds[:1])                          # first use: workspace, depth table
    t0 = time.perf_counter()
    try:
        lik = api.solve_stage(aa, every, ex, 1, seeds)[2][:, 0]
    except Exception as e:                                                     # LCTY_ERR_SOLVER: a genotype without an answer inside the node limit
        return {"workload": "10000 read pairs x 8 alleles (BASELINE.json configs[0]), 36 genotypes x 1 attempt", "error": str(e)}, None
    wall = time.perf_counter() - t0
    chains = {}
    for name, kind in (("greedy", cdefs.SOLVER_GREEDY), ("annealing", cdefs.SOLVER_ANNEAL)):
        cl = api.solve_stage(aa, every, api.default_solver(kind), 1, seeds)[2][:, 0]
        chains[name + "_chain_below_exact_max"] = float(np.max(lik - cl))
        chains[name + "_chains_above_exact"] = int(np.count_nonzero(cl > lik + 1e-9 * np.abs(lik)))
    leg = {"workload": "10000 read pairs x 8 alleles (BASELINE.json configs[0]): all 36 genotypes x 1 attempt, relative gap 1e-4 (HiGHS' default)",
           "genotypes": int(len(every)), "answered": int(np.count_nonzero(np.isfinite(lik))), "seconds": wall, "genotypes_per_s": len(every) / wall,
           "host_threads": "the pool of the library (exact_threads: the machine's hardware threads, at most 64)", **chains}
    return leg, (L, p, loc, aa, every, seeds, lik)
