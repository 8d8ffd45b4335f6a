"""Developer measurement: lcty_recover_alignments on a batch of synthetic Illumina read pairs x 256 alleles, primary records only (the bench's
`recovery` leg), once per knob setting:  python3 scripts/recover_probe.py [--diag] [--lib path] [pairs] [name=value[,name=value] ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locityper_amd import api, synth, _lib

def main():
    args = sys.argv[1:]
    if "--lib" in args:
        _lib.LIB_PATH = os.path.abspath(args[args.index("--lib") + 1]); del args[args.index("--lib"):args.index("--lib") + 2]
    if "--diag" in args:
        _lib.use_diag_build(); args.remove("--diag")
    n = int(args[0]) if args and args[0].isdigit() else 262144
    settings = [a for a in args if "=" in a] or ["default"]
    A = 256
    ctx = api.Context(0)
    L = synth.SynthLocus(A, n, seed=synth.SEED)
    p = api.resolve_params(api.default_params(), L.bg)
    loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    loc.set_hap_alns(L.hap_alns(), transfer_fails=100, max_div=0.1)
    prim = [L.reads(lo, min(32768, n - lo), primaries_only=True) for lo in range(0, n, 32768)]
    ab = api.AllAlignments.load(loc, prim); ab.recover(); ab.close()              # the context's scratch
    for st in settings:
        knobs = [] if st == "default" else [kv.split("=") for kv in st.split(",")]
        for k, v in knobs: ctx.set_knob(k, int(v))
        ab = api.AllAlignments.load(loc, prim)
        ctx.timing_reset()
        t = time.perf_counter()
        m = ab.recover()
        dt = time.perf_counter() - t
        nk, ms = ctx.timing(api.K_TRANSFER)
        print(f"{st}: wall {dt:.3f} s, transfer kernel {ms:.1f} ms x{nk}, new {m} ({m / max(ms, 1e-9) / 1e3:.1f} M transfers/s)", flush=True)
        ab.close()
        for k, _ in knobs: ctx.set_knob(k, -1)

main()
