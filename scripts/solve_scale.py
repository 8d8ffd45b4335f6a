"""Developer measurement: solver-stage kernel time vs read count / chain count (config-2 shaped locus)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from locityper_amd import api, synth, cdefs, _lib
if os.environ.get("LCTY_EXPERIMENT_LIB"):              # developer experiments only (a variant library built by hand)
    _lib.LIB_PATH = os.environ["LCTY_EXPERIMENT_LIB"]

def main():
    A = int(sys.argv[1]); pairs = int(sys.argv[2]); chains = [int(x) for x in sys.argv[3].split(",")]
    kinds = sys.argv[4] if len(sys.argv) > 4 else "ga"
    L = synth.SynthLocus(A, pairs, seed=synth.SEED)
    p = api.resolve_params(api.default_params(), L.bg)
    ctx = api.Context(0)
    loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    aa = None
    chunk = 32768
    for lo in range(0, pairs, chunk):
        ch = L.reads(lo, min(chunk, pairs - lo))
        if aa is None:
            f = 1.05 * pairs / ch.n_pairs
            aa = api.AllAlignments(loc, pairs, (int(ch.n_bases * f) + 2048) // 32 * 32, int(len(ch.recs) * f) + 4096, int(len(ch.cigar) * f) + 65536)
        aa.append(ch)
    aa.score()
    sc = aa.run_filter()
    gts = api.generate_genotypes(A, 2)
    order = np.argsort(-sc, kind="stable")
    print(f"A={A} pairs={pairs} n_good={aa.n_good()} max n_windows={max(loc.contig_info(a)[3] for a in range(min(A,4)))}", flush=True)
    cpws = [int(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else [0]
    if len(sys.argv) > 6: ctx.set_knob("solve_lds_weights", int(sys.argv[6]))
    ctx.set_knob("solve_stats", 1)
    api.solve_stage(aa, gts[order[:max(chains)]], api.default_solver(cdefs.SOLVER_GREEDY), 1, api.chain_seeds(7, max(chains)))   # warm-up: tables, workspace
    for kind, name in ((cdefs.SOLVER_GREEDY, "g"), (cdefs.SOLVER_ANNEAL, "a")):
        if name not in kinds: continue
        for cpw in (cpws if name == "g" else [0]):
            ctx.set_knob("solve_chains_per_wave", cpw if cpw else -1)
            for n in chains:
                sub = gts[order[:n]]
                ctx.timing_reset()
                t = time.time()
                m, v, l = api.solve_stage(aa, sub, api.default_solver(kind), 1, api.chain_seeds(7, n))
                dt = time.time() - t
                ks = ctx.timing(api.K_SOLVE if name == "g" else api.K_ANNEAL)[1]
                print(f"  kind={name} cpw={cpw} chains={n}: wall {dt:.3f} s loop kernel {ks:.1f} ms init {ctx.timing(api.K_SOLVE_INIT)[1]:.1f} ms "
                      f"best={sub[int(np.argmax(m))]} true={L.true_genotype}", flush=True)

main()
