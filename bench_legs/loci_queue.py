"""The queue of loci as `locityper genotype` meets it: every locus arrives from the host (lcty_solve_queue_fed)."""
import threading
import time

import numpy as np

from locityper_amd import api, synth
from locityper_amd.cdefs import ALN_REC_DTYPE, ReadsChunk
from .common import genotype_of, progress
from .rooflines import KERNEL_IDS


def distinct_loci_leg(args, ctx, loci, batches, stages, gts, resident_ms_per_step, host_chunks):
    """D distinct loci — the loci of the main measurement first, so that with the default D = 2 the two queues do the same work — (their
    counted alignment tables and bases in page-locked memory: 8.3 GB each at 1 M x 256), K positions cycling over them; a loader thread
    resets one of three batch objects (lcty_reads_reset), uploads the position's chunks (lcty_reads_append_counted: copies on the
    context's copy stream, the CSR validation on the host's cores next to them) and hands it to lcty_solve_queue_fed, which releases a
    batch when its last stage is done. Timed: K positions, the first upload included."""
    D, K, A = args.distinct_loci, (args.distinct_steps or args.steps), args.alleles
    trace = any(kv.startswith("queue_trace=") and not kv.endswith("=0") for kv in args.knob)
    for b in batches:                                          # the resident loci of the main measurement make room
        b.close()
    ctx.trim()
    t0 = time.time()
    host = []                                                  # per locus: (SynthLocus, Locus, [(pinned chunk, pinned counted alignments)])
    n_chunks = (args.pairs + args.chunk - 1) // args.chunk
    none_recs, none_cig = np.zeros(0, dtype=ALN_REC_DTYPE), np.zeros(0, dtype=np.uint32)
    caps = None
    up_bytes = 0
    for j in range(D):
        if j < len(loci):
            L, loc = loci[j]
        else:
            L = synth.SynthLocus(A, args.pairs, seed=synth.SEED + 100 + j)
            loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, api.resolve_params(api.default_params(), L.bg))
        chunks = []
        tb = tr = 0
        for ci in range(n_chunks):
            lo = ci * args.chunk
            kept = j < len(host_chunks) and host_chunks[j]
            ch = host_chunks[j][ci] if kept else L.reads(lo, min(args.chunk, args.pairs - lo))
            alns = ctx.pinned_like(ch.counted(loc.allele_len))
            pinned = [ctx.pinned_like(a) for a in (ch.mate_len, ch.mate_off, ch.bases2, ch.nmask, ch.aln_off)]
            pc = ReadsChunk(*pinned, none_recs, np.zeros(ch.n_pairs + 1, dtype=np.uint64), none_cig)
            chunks.append((pc, alns))
            tb += ch.n_bases
            tr += len(ch.recs)
            if j == 0:
                up_bytes += alns.nbytes + sum(a.nbytes for a in pinned)
            if kept:
                host_chunks[j][ci] = None
            del ch
        caps = (max(caps[0], tb), max(caps[1], tr)) if caps else (tb, tr)
        host.append((L, loc, chunks))
    ctx.set_knob("arena_cap_pct", 35)                         # one PairAlignment per (pair, allele) is the rule here; the bound is two per record
    cap_bases = (int(caps[0] * 1.01) + 1024) // 32 * 32 + 32
    rot = [api.AllAlignments(host[0][1], args.pairs, cap_bases, int(caps[1] * 1.01) + 4096, 0) for _ in range(3)]
    ctx.set_knob("arena_cap_pct", -1)
    setup_s = time.time() - t0
    loaded_once = [False]

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
                    _, loc_i, chunks_i = host[(first_it + i) % D]
                    b = rot[i % 3]
                    if not (args.distinct_no_upload and first_it == 0 and loaded_once[0]):
                        b.reset(loc_i)
                        for pc, alns in chunks_i:
                            b.append(pc, counted=alns)
                    load_s[i] = time.perf_counter() - tl
                    if trace:
                        progress(f"  position {i}: loaded in {load_s[i]:.3f} s")
                    ready[i].set()
            except BaseException as e:                         # the queue must not wait for ever
                problems.append(e)
                for ev in ready:
                    ev.set()

        def acquire(i):
            tw = time.perf_counter()
            ready[i].wait()
            if trace:
                progress(f"  position {i}: acquired after waiting {time.perf_counter() - tw:.3f} s")
            if problems:
                raise problems[0]
            return rot[i % 3]

        def release(i):
            if trace:
                progress(f"  position {i}: released")
            free[i % 3].release()

        th = threading.Thread(target=loader, daemon=True)      # a queue that raised must not leave the process waiting for its loader
        ctx.synchronize()
        tq = time.perf_counter()
        th.start()
        try:
            calls = api.solve_queue_fed(k, acquire, release, stages, master_seeds=[3000 + first_it + i for i in range(k)])
        except BaseException:
            for f in free:
                f.release()                                     # the loader may sit in an acquire: let it run out
            raise
        ctx.synchronize()
        dt = time.perf_counter() - tq
        th.join()
        ok = all(genotype_of(gts, c.ixs[0]) == tuple(host[(first_it + i) % D][0].true_genotype) for i, c in enumerate(calls))
        return dt, ok, load_s

    run(3, 0)                                                 # every batch object once: workspaces, page tables
    loaded_once[0] = True
    ctx.timing_reset()
    dt, ok, load_s = run(K, 0 if args.distinct_no_upload else 1)
    kern = {name: ctx.timing(getattr(api, kid))[1] / K for name, kid in KERNEL_IDS if name != "solve_init_kernel_annealing_stage"}
    for b in rot:
        b.close()
    ms = 1e3 * dt / K
    return {"what": f"{K} positions over {D} distinct loci of {args.pairs} read pairs x {A} alleles; every position uploaded from page-locked "
                    "host memory (lcty_reads_reset + lcty_reads_append_counted from a loader thread, copy stream) while the position before "
                    "it is solved (lcty_solve_queue_fed, three batch objects); the first upload of the queue is inside the timed region",
            "ms_per_step": ms, "read_pairs_per_s": args.pairs * K / dt, "resident_ms_per_step": resident_ms_per_step,
            "ratio_to_resident": ms / resident_ms_per_step, "all_calls_equal_truth": ok, "kernel_ms_per_step": kern,
            "upload_GB_per_locus": up_bytes / 1e9, "upload_and_validate_s_per_locus": float(np.median(load_s)),
            "upload_GBs": up_bytes / 1e9 / float(np.median(load_s)), "setup_s": setup_s}
