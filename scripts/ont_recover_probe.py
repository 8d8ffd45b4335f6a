"""Developer measurement: lcty_recover_alignments on a batch of synthetic 10-kb ONT reads x 256 alleles (wall vs kernel time), once per
knob setting:  python3 scripts/ont_recover_probe.py [--diag] [reads] [name=value[,name=value] ...]; the products of every setting must equal
those of the first (statuses, matrix)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locityper_amd import api, synth, cdefs

def main():
    import numpy as np
    args = sys.argv[1:]
    if "--lib" in args:
        from locityper_amd import _lib
        _lib.LIB_PATH = os.path.abspath(args[args.index("--lib") + 1]); del args[args.index("--lib"):args.index("--lib") + 2]
    if "--diag" in args:                                          # the developer build: knob transfer_phases exists there only
        from locityper_amd import _lib
        _lib.use_diag_build(); args.remove("--diag")
    nont = int(args[0]) if args and args[0].isdigit() else 6144
    settings = [a for a in args if "=" in a] or ["default"]
    A = 256
    ctx = api.Context(0)
    Lo = synth.SynthLocus(A, nont, seed=synth.SEED + 77, technology=cdefs.TECH_NANOPORE, read_len=10_000)
    po = api.resolve_params(api.default_params(), Lo.bg)
    loco = api.Locus(ctx, Lo.seqs, Lo.seq_off, Lo.counts, Lo.cnt_off, Lo.k, Lo.bg, po)
    loco.set_hap_alns(Lo.hap_alns(), transfer_fails=100, max_div=0.1)
    prim = [Lo.reads(lo, min(256, nont - lo), primaries_only=True) for lo in range(0, nont, 256)]
    ao = api.AllAlignments.load(loco, prim); ao.recover(); ao.close()          # the context's scratch
    first = None
    for st in settings:
        knobs = [] if st == "default" else [kv.split("=") for kv in st.split(",")]
        for k, v in knobs: ctx.set_knob(k, int(v))
        ao = api.AllAlignments.load(loco, prim)
        ctx.timing_reset()
        t = time.perf_counter()
        n = ao.recover()
        dt = time.perf_counter() - t
        prod = (ao.status()[0].tobytes(), ao.best_aln_matrix().tobytes())
        if first is None: first = prod
        print(f"{st}: wall {dt:.3f} s, transfer kernel {ctx.timing(api.K_TRANSFER)[1]:.1f} ms x{ctx.timing(api.K_TRANSFER)[0]}, score {ctx.timing(api.K_SCORE)[1]:.1f} ms, "
              f"new {n} ({n / max(ctx.timing(api.K_TRANSFER)[1], 1e-9) / 1e3:.2f} M transfers/s), products equal the first setting's: {prod == first}", flush=True)
        ao.close()
        for k, _ in knobs: ctx.set_knob(k, -1)

main()
