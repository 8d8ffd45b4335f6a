"""Developer measurement: the exact solver at its default gap on the best N genotypes of configs[0] (10 000 read pairs x 8 alleles):
   python3 scripts/exact_stage_probe.py [n_genotypes] [node_limit]     root gap, free reads, nodes and seconds per genotype (trace on)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from locityper_amd import _lib, api, synth, cdefs

n_gt = int(sys.argv[1]) if len(sys.argv) > 1 else 20
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
ctx = api.Context(0)
L = synth.SynthLocus(8, 10_000, seed=synth.SEED + 3)
p = api.resolve_params(api.default_params(), L.bg)
loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
aa = api.AllAlignments.load(loc, L.reads(0, 10_000))
gts = api.generate_genotypes(8, 2)
sub = np.ascontiguousarray(gts[np.argsort(-aa.run_filter(), kind="stable")[:n_gt]])
seeds = api.chain_seeds(5, len(sub))
ex = api.default_solver(cdefs.SOLVER_EXACT)
ex.node_limit = limit
ctx.set_knob("exact_trace", 1)
ctx.set_knob("exact_threads", 1)
g = api.solve_stage(aa, sub, api.default_solver(cdefs.SOLVER_GREEDY), 1, seeds)[2][:, 0]
a = api.solve_stage(aa, sub, api.default_solver(cdefs.SOLVER_ANNEAL), 1, seeds)[2][:, 0]
for gi in range(len(sub)):
    t0 = time.perf_counter()
    try:
        e = api.solve_stage(aa, sub[gi:gi + 1], ex, 1, seeds[gi:gi + 1])[2][0, 0]
        msg = f"answer {e:.4f} (greedy {g[gi] - e:+.3e}, annealing {a[gi] - e:+.3e})"
    except _lib.LocityperError as err:
        msg = "refused: " + str(err)[:90]
    print(f"genotype {tuple(int(x) for x in sub[gi])} (true {tuple(L.true_genotype)}): {msg}; {time.perf_counter() - t0:.2f} s", flush=True)
