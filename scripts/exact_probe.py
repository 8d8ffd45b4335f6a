"""Developer measurement: the exact solver (LCTY_SOLVER_EXACT) on loci of growing size — BASELINE configs[0] is 10 000 read pairs x 8 alleles.
   python3 scripts/exact_probe.py [pairs ...]     For the true genotype and its best competitor: proven optimum vs the chains, nodes, seconds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from locityper_amd import _lib
_lib.use_diag_build()                                        # exact_trace is a knob of the developer build (make -C locityper_amd/csrc DIAG=1)
from locityper_amd import api, synth, cdefs


def main():
    gap = None                                   # the library's default: 1e-4, HiGHS' mip_rel_gap
    if "--gap" in sys.argv:
        gap = float(sys.argv[sys.argv.index("--gap") + 1]); del sys.argv[sys.argv.index("--gap"):sys.argv.index("--gap") + 2]
    dump = "--dump" in sys.argv
    if dump:
        sys.argv.remove("--dump")
        os.makedirs("gpurun_out", exist_ok=True)
    sizes = [int(a) for a in sys.argv[1:]] or [100, 300, 1000, 3000, 10000]
    ctx = api.Context(0)
    ctx.set_knob("exact_trace", 1)                            # (the developer build: _lib.use_diag_build() above)
    if dump: ctx.set_path("exact_dump", "gpurun_out/exact_model.txt")
    for n in sizes:
        L = synth.SynthLocus(8, n, seed=synth.SEED + 3)
        p = api.resolve_params(api.default_params(), L.bg)
        loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
        aa = api.AllAlignments.load(loc, L.reads(0, n))
        gts = api.generate_genotypes(8, 2)
        order = np.argsort(-aa.run_filter(), kind="stable")[:2]
        sub = np.ascontiguousarray(gts[order])
        seeds = api.chain_seeds(5, len(sub))
        g = api.solve_stage(aa, sub, api.default_solver(cdefs.SOLVER_GREEDY), 1, seeds)[2][:, 0]
        a = api.solve_stage(aa, sub, api.default_solver(cdefs.SOLVER_ANNEAL), 1, seeds)[2][:, 0]
        ex = api.default_solver(cdefs.SOLVER_EXACT)
        if gap is not None: ex.init_prob = gap
        for gi in range(len(sub)):
            t0 = time.perf_counter()
            try:
                e = api.solve_stage(aa, sub[gi:gi + 1], ex, 1, seeds[gi:gi + 1])[2][0, 0]
                msg = f"optimum {e:.6f} (greedy {g[gi] - e:+.3e}, annealing {a[gi] - e:+.3e})"
            except _lib.LocityperError as err:
                msg = f"refused: {err}"
            print(f"{n} read pairs, genotype {tuple(int(x) for x in sub[gi])}: {msg}; {time.perf_counter() - t0:.2f} s", flush=True)


main()
