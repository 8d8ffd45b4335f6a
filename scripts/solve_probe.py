"""Developer measurement: the solver stages of BASELINE configs[1] (1 M read pairs x 256 alleles) ALONE, once per knob setting:
   python3 scripts/solve_probe.py [--diag] [--chains 5000] [--short] [--base-len 50000] name=value[,name=value...] ...
For every setting: solve_init_kernel and greedy_loop_kernel times of the default greedy stage (5 000 chains, 100 000 iterations;
--short: plateau 1, the stage is its initialisation) and whether the per-chain likelihoods equal those of the first setting bit for bit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from locityper_amd import api, synth, cdefs


def main():
    args = [a for a in sys.argv[1:]]
    n = 5000
    short = False
    anneal = False                                    # --anneal: the annealing stage of the default scheme instead (20 genotypes x 20 attempts)
    settings = []
    sort_gts = False
    base_len = 50_000                                 # --base-len: shorter alleles = fewer windows per chain = more greedy wavefronts per CU
    i = 0
    while i < len(args):
        if args[i] == "--chains": n = int(args[i + 1]); i += 2
        elif args[i] == "--lib":
            from locityper_amd import _lib
            _lib.LIB_PATH = os.path.abspath(args[i + 1]); i += 2          # a variant library built by hand (hipcc over a patched copy of a source)
        elif args[i] == "--diag":                                 # the developer build: knobs such as solve_stats, solve_greedy_form, solve_init_tiles exist there only
            from locityper_amd import _lib
            _lib.use_diag_build(); i += 1
        elif args[i] == "--short": short = True; i += 1
        elif args[i] == "--base-len": base_len = int(args[i + 1]); i += 2
        elif args[i] == "--sort-genotypes": sort_gts = True; i += 1      # the stage's genotypes in lexicographic order instead of by prefilter score
        elif args[i] == "--anneal": anneal = True; i += 1
        else: settings.append(args[i]); i += 1
    if not settings: settings = ["default"]
    A, pairs = 256, 1_000_000
    L = synth.SynthLocus(A, pairs, seed=synth.SEED, base_len=base_len)
    p = api.resolve_params(api.default_params(), L.bg)
    ctx = api.Context(0)
    loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    aa = None
    for lo in range(0, pairs, 32768):
        ch = L.reads(lo, min(32768, pairs - lo))
        if aa is None:
            f = 1.05 * pairs / ch.n_pairs
            aa = api.AllAlignments(loc, pairs, (int(ch.n_bases * f) + 2048) // 32 * 32, int(len(ch.recs) * f) + 4096, 0)
        aa.append(ch, counted=True)
    aa.score()
    sc = aa.run_filter()
    gts = api.generate_genotypes(A, 2)
    order = np.argsort(-sc, kind="stable")
    top = np.ascontiguousarray(gts[order[:n]])
    if sort_gts: top = np.ascontiguousarray(top[np.lexsort((top[:, 1], top[:, 0]))])
    sv = api.default_solver(cdefs.SOLVER_ANNEAL if anneal else cdefs.SOLVER_GREEDY)
    if short: sv.plato_size = 1
    att = 1
    if anneal: top, att, n = np.ascontiguousarray(top[:20]), 20, 400
    seeds = api.chain_seeds(7, n)
    api.solve_stage(aa, top[:min(64, len(top))], sv, att, seeds[:min(64, len(top)) * att])          # allocations
    first = None
    for st in settings:
        knobs = [] if st == "default" else [kv.split("=") for kv in st.split(",")]
        for k, v in knobs: ctx.set_knob(k, int(v))
        api.solve_stage(aa, top, sv, att, seeds)              # warm (workspace growth)
        ctx.timing_reset()
        t0 = time.perf_counter()
        m, v_, l = api.solve_stage(aa, top, sv, att, seeds)
        wall = time.perf_counter() - t0
        if first is None: first = l
        ch_, it_, acc_ = api.solve_stats(aa)
        print(f"{st}: {ch_} chains, {it_} iterations, {acc_} accepted; init {ctx.timing(api.K_SOLVE_INIT)[1]:.1f} ms, loop {ctx.timing(api.K_ANNEAL if anneal else api.K_SOLVE)[1]:.1f} ms, wall {1e3 * wall:.1f} ms, "
              f"likelihoods equal the first setting's: {bool(np.array_equal(l, first))} (largest relative difference {float(np.max(np.abs(l - first) / np.abs(first))):.3g})", flush=True)
        for k, _ in knobs: ctx.set_knob(k, -1)


main()
