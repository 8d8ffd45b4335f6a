"""Developer measurement: BASELINE configs[2] (single-end 10 kb ONT reads x 256 alleles) through a STREAMING batch — the records
of one chunk on the device at a time (600 KB of CIGAR words per read: 1 M reads = 600 GB) — then prefilter and the default
solver scheme on the kept products. Reports the kernel rate (chunk resident) and the end-to-end rate including host
generation / PCIe upload. Parity of the streaming path: tests/test_gpu_streaming.py.
usage: ont_stream_probe.py [reads] [alleles] [chunk]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from locityper_amd import api, synth, cdefs


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    A = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
    pinned = len(sys.argv) > 4 and sys.argv[4] == "pinned"            # chunks in page-locked memory (lcty_host_alloc)
    L = synth.SynthLocus(A, n, technology=cdefs.TECH_NANOPORE, read_len=10_000)
    p = api.resolve_params(api.default_params(), L.bg)
    ctx = api.Context(0)
    loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    c0 = L.reads(0, min(chunk, n))
    est = lambda x: int(x * 1.1) + 65536
    aa = api.AllAlignments(loc, n, est(c0.n_bases) // 32 * 32 + 32, est(len(c0.recs)), est(len(c0.cigar)), streaming_chunk_pairs=chunk)
    t_gen = t_up = t_score = 0.0
    recs = words = 0
    ctx.timing_reset()
    t_all = time.perf_counter()
    for lo in range(0, n, chunk):
        t0 = time.perf_counter()
        ch = c0 if lo == 0 else L.reads(lo, min(chunk, n - lo))
        if pinned: ch = ctx.pinned_chunk(ch)
        t1 = time.perf_counter()
        aa.append(ch)
        t2 = time.perf_counter()
        aa.score(); ctx.synchronize()
        t3 = time.perf_counter()
        t_gen += t1 - t0; t_up += t2 - t1; t_score += t3 - t2
        recs += len(ch.recs); words += len(ch.cigar)
        del ch
    t_load = time.perf_counter() - t_all
    k, ms = ctx.timing(api.K_SCORE)
    t0 = time.perf_counter()
    scores = aa.run_filter()
    gts = api.generate_genotypes(A, 2)
    keep = api.truncate_ixs(scores, np.arange(len(scores), dtype=np.uint64), p.filt_diff, 5000, 1)
    t_pref = time.perf_counter() - t0
    t0 = time.perf_counter()
    greedy, anneal = api.default_solver(cdefs.SOLVER_GREEDY), api.default_solver(cdefs.SOLVER_ANNEAL)
    mean, var, att = np.full(len(gts), np.nan), np.full(len(gts), np.nan), np.zeros(len(gts), dtype=np.uint32)
    m, v, _ = api.solve_stage(aa, gts[keep], greedy, 1, api.chain_seeds(1, len(keep)))
    mean[keep], var[keep], att[keep] = m, v, 1
    ixs = api.discard_improbable(mean, var, att, keep, p.prob_thresh, 20, 1)
    m, v, _ = api.solve_stage(aa, gts[ixs], anneal, 20, api.chain_seeds(2, 20 * len(ixs)))
    mean[ixs], var[ixs], att[ixs] = m, v, 20
    res = api.produce_result(mean, var, att, ixs, p.prob_thresh)
    t_solve = time.perf_counter() - t0
    out = {"workload": f"{n} single-end 10 kb ONT reads x {A} alleles, streaming batch with chunks of {chunk} reads", "host_memory": "page-locked" if pinned else "pageable",
           "records": recs, "cigar_words": words, "raw_GB": (16 * recs + 4 * words) / 1e9,
           "score_kernel_ms_total": ms, "reads_per_s_kernel": n / ms * 1e3, "records_cigar_GBs_kernel": (16 * recs + 4 * words) / ms / 1e6,
           "host_generation_s": t_gen, "upload_s": t_up, "score_s": t_score, "load_s_end_to_end": t_load,
           "reads_per_s_end_to_end_incl_generation_and_pcie": n / t_load, "upload_GBs": (16 * recs + 4 * words) / 1e9 / max(t_up, 1e-9),
           "prefilter_s": t_pref, "solver_scheme_s": t_solve, "good_reads": aa.n_good(),
           "called": [int(x) for x in gts[int(res[0][0])]], "true": list(L.true_genotype)}
    print(json.dumps(out), flush=True)


main()
