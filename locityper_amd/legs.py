"""Measurement legs shared by bench.py and scripts/ (harness code: ctypes calls into liblocityper_hip.so, nothing computed here).

ont_from_bases_on_a_basis: BASELINE.json configs[2] — 10-kb ONT reads x 256 alleles — from the bases alone, the way `locityper genotype
--basis` does it (src/command/genotype.rs:1007-1050: the mapper sees a BASIS of the locus' haplotypes, the other alleles are reached by
HapAlns::transfer_alignments, seq/transfer.rs:70-140): per chunk of a streaming batch
    lcty_reads_map_append   candidate generation, long route, onto the basis alleles (seeds -> chains -> banded gap-affine alignment)
    lcty_score_reads        AllAlignments::load, first pass
    lcty_recover_alignments every other allele through the haplotype-to-haplotype alignments
    lcty_score_reads        second pass
and lcty_prefilter over all genotypes at the end."""
import time

import numpy as np

from . import api, cdefs, synth


def choose_basis(hap_alns, n_alleles, n_basis):
    """A basis the way `locityper augment` makes one: alleles picked so that every allele has a CLOSE member in the basis (greedy k-centre on
    the divergences of the haplotype-to-haplotype alignments, (aln_len - n_matches) / aln_len): alignment transfer is exact where the
    two alleles agree, and a read that comes from allele t is first taken from the basis allele nearest to t."""
    D = np.zeros((n_alleles, n_alleles))
    for q, r, _, nm, ln in hap_alns:
        D[q, r] = D[r, q] = (ln - nm) / max(ln, 1)
    basis = [int(np.argmin(D.sum(axis=1)))]                         # the medoid first
    near = D[basis[0]].copy()
    while len(basis) < min(n_basis, n_alleles):
        nxt = int(np.argmax(near))
        basis.append(nxt)
        near = np.minimum(near, D[nxt])
    return sorted(basis), float(near.max())


def ont_from_bases_on_a_basis(ctx, n_reads, n_alleles=256, n_basis=16, chunk=2048, read_len=10_000, seed_off=77, progress=None):
    say = progress or (lambda *_: None)
    A = n_alleles
    t0 = time.perf_counter()
    L = synth.SynthLocus(A, n_reads, seed=synth.SEED + seed_off, technology=cdefs.TECH_NANOPORE, read_len=read_len)
    p = api.resolve_params(api.default_params(), L.bg)
    loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    H = L.hap_alns()
    loc.set_hap_alns(H, transfer_fails=100, max_div=0.1)
    t_locus = time.perf_counter() - t0
    mp = api.map_params(long_reads=True)
    basis, worst = choose_basis(H, A, n_basis)
    t0 = time.perf_counter()
    api.build_map_index(loc, basis, k=mp.k)
    t_index = time.perf_counter() - t0
    say(f"  locus + {len(H)} haplotype alignments {t_locus:.1f} s, index over {len(basis)} basis alleles {t_index:.2f} s")
    # the chunks as the sequencer gave them (no records), generated ahead: the generator is not what is measured
    chunks = [synth.sequencer_orientation(L.reads(lo, min(chunk, n_reads - lo), primaries_only=True)) for lo in range(0, n_reads, chunk)]
    cb = max(int(c.n_bases) for c in chunks); rb = max(int(c.mate_len.sum()) for c in chunks)
    read_bases = sum(int(c.mate_len.sum()) for c in chunks)

    def batch():
        return api.AllAlignments(loc, n_reads, (cb + 2048) // 32 * 32, chunk * len(basis) * 2 + 1024, rb // 3 * len(basis) + 4096,
                                 streaming_chunk_pairs=chunk)
    # first use: code objects, the mapper's buffers, the lane scratch of the recovery (tens of GB: seconds of allocation, once per context)
    aw = batch()
    api.map_append(aw, chunks[0], mp); aw.score(); aw.recover()
    aw.close()
    aa = batch()
    ctx.synchronize()
    ctx.timing_reset()
    t_map = t_score = t_rec = 0.0
    n_new = 0
    t0 = time.perf_counter()
    for c in chunks:
        t1 = time.perf_counter()
        api.map_append(aa, c, mp)
        t2 = time.perf_counter()
        aa.score()
        t3 = time.perf_counter()
        n_new += aa.recover()                               # with its second scoring pass
        t4 = time.perf_counter()
        t_map += t2 - t1; t_score += t3 - t2; t_rec += t4 - t3
    ctx.synchronize()
    t_stream = time.perf_counter() - t0
    t0 = time.perf_counter()
    sc = aa.run_filter()
    t_filter = time.perf_counter() - t0
    n_map, ms_map = ctx.timing(api.K_MAP)
    n_tr, ms_tr = ctx.timing(api.K_TRANSFER)
    _, ms_score = ctx.timing(api.K_SCORE)
    gts = api.generate_genotypes(A, 2)
    best = tuple(int(x) for x in gts[int(np.argmax(sc))])
    truth = tuple(int(x) for x in L.true_genotype)
    truth_ix = next(i for i, g in enumerate(gts) if tuple(int(x) for x in g) == truth)
    mean_len = read_bases / max(n_reads, 1)
    band = 2 * mp.band + 1
    cells = float(n_reads) * len(basis) * mean_len * band          # every basis allele is aligned in a band of +-band around its chain
    total = t_stream + t_filter
    out = {
        "sample": f"{n_reads} synthetic {read_len}-base ONT reads as sequenced (no records) x {A} alleles, streamed in chunks of {chunk}: mapped onto "
                  f"{len(basis)} basis alleles on the device (seeds of {mp.k} every {mp.stride} bases, one chain per (allele, strand), gap-affine alignment in a "
                  f"band of +-{mp.band}), the other {A - len(basis)} alleles through {len(H)} haplotype alignments (transfer_fails 100), scored, prefiltered "
                  f"over all {len(gts)} genotypes — the reference's `--basis` route (genotype.rs:1007-1050)",
        "reads": n_reads, "alleles": A, "basis_alleles": len(basis), "chunk": chunk,
        "basis": "greedy k-centre on the divergences of the haplotype alignments (as `locityper augment` picks a basis)", "largest_divergence_to_the_basis": worst,
        "reads_per_s_bases_to_prefilter": n_reads / total, "seconds": total,
        "map_call_s": t_map, "score_call_s": t_score, "recover_and_rescore_call_s": t_rec, "run_filter_s": t_filter,
        "reads_per_s_map_call": n_reads / t_map if t_map else None,
        "map_kernels_ms": ms_map, "transfer_kernel_ms": ms_tr, "score_kernels_ms": ms_score,
        "alignments_mapped": n_reads * len(basis), "alignments_transferred": int(n_new),
        "transfers_per_s_kernel": n_new / (ms_tr * 1e-3) if ms_tr else None,
        "alignments_per_s_map_kernels": n_reads * len(basis) / (ms_map * 1e-3) if ms_map else None,
        "good_reads": aa.n_good(), "locus_and_hap_alns_s": t_locus, "index_build_s": t_index,
        "best_genotype": list(best), "true_genotype": list(truth), "prefilter_best_is_truth": best == truth,
        "truth_scores_as_the_best": bool(sc[truth_ix] >= sc.max() - 1e-9 * abs(sc.max())),
        # the dominant kernel of this leg is the band DP of the mapper: integer VALU work, not memory
        "roofline": {"bound": "valu_int", "kernel": "map_long_align_kernel", "unit": "G band cells/s",
                     "achieved": cells / (ms_map * 1e-3) / 1e9 if ms_map else None,
                     # 1 024 SIMDs x 16 lanes x 2.4 GHz = 39.3 T integer lane-operations/s; a cell of the gap-affine recurrence with its
                     # five direction bits is ~12 of them (three maxima with their sources, the base comparison, two gap updates)
                     "peak": 39.3e3 / 12.0, "frac": (cells / (ms_map * 1e-3) / 1e9) / (39.3e3 / 12.0) if ms_map else None,
                     "what": "nodes of the band of the gap-affine alignment (read bases x basis alleles x (2 * band + 1)) per second of the mapper's kernels, "
                             "against the integer VALU rate at ~12 lane-operations per node; HBM traffic is small beside it (profiles/r03_pmc_map_long_*)",
                     "traffic": None},
    }
    aa.close()
    return out, (L, loc, chunks)


def ont_from_bases_stream(ctx, n_reads, n_alleles=256, chunk=1024, read_len=10_000, seed_off=77, progress=None):
    """BASELINE.json configs[2] as named — 10-kb ONT reads x 256 alleles, the long-read DP path — from the bases alone to the prefilter call,
    through a streaming batch: every chunk of reads as sequenced is mapped onto EVERY allele on the device (long route: seeds -> one chain
    per (allele, strand) -> banded gap-affine alignment; the reference runs minimap2 -N min(25 000, 4 x alleles) here, genotype.rs:990-1002),
    its records go straight into the batch and are scored (AllAlignments::load), then dropped; run_filter over all genotypes at the end.
    (Mapping onto a basis of 16 alleles and reaching the others by alignment transfer is three times as fast and calls the wrong genotype
    on these reads: ont_from_bases_on_a_basis and DESIGN.md say why.) Returns the leg's figures and (locus generator, resolved parameters)
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
        "sample": f"{n_reads} synthetic {read_len}-base ONT reads as sequenced (no records) x {A} alleles (BASELINE.json configs[2]), streamed in chunks of "
                  f"{chunk}: mapped onto all {A} alleles on the device (seeds of {mp.k} every {mp.stride} bases, one chain per (allele, strand), gap-affine "
                  f"alignment in a band of +-{mp.band}), records straight into a streaming batch, scored chunk by chunk, prefiltered over all {len(gts)} genotypes",
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
                     "what": "nodes of the band of the gap-affine alignment (read bases x alleles x (2 * band + 1)) per second of the mapper's kernels against "
                             "the integer VALU rate (1 024 SIMDs x 16 lanes x 2.4 GHz) at ~12 lane-operations per node; a band of +-16 keeps 33 of 64 lanes "
                             "busy and a row costs ~40 vector + ~22 scalar instructions (profiles/r03_pmc_map_long_2048_ont_reads_x16.json); HBM traffic is "
                             "~120 KB per alignment, small beside it"},
    }
    aa.close()
    return out, (L, p)
