"""Developer measurement: the scoring kernel of BASELINE configs[1] (1 M read pairs x 256 alleles) alone.
   python3 scripts/score_probe.py [--lib variant.so] [--format counted|records] [--pairs N] [knob=value ...]
Prints the kernel time of lcty_score_reads (mean of 5 launches) and a checksum of its products (statuses, matrix sum, arena size)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    args = sys.argv[1:]
    fmt, pairs, knobs = "counted", 1_000_000, []
    i = 0
    while i < len(args):
        if args[i] == "--lib":
            from locityper_amd import _lib
            _lib.LIB_PATH = os.path.abspath(args[i + 1]); i += 2
        elif args[i] == "--format": fmt = args[i + 1]; i += 2
        elif args[i] == "--pairs": pairs = int(args[i + 1]); i += 2
        elif "=" in args[i]: knobs.append(args[i].split("=")); i += 1
        else: raise SystemExit("unknown argument " + args[i])
    from locityper_amd import api, synth, cdefs
    A = 256
    L = synth.SynthLocus(A, pairs, seed=synth.SEED)
    p = api.resolve_params(api.default_params(), L.bg)
    ctx = api.Context(0)
    for name, val in knobs: ctx.set_knob(name, int(val))
    loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    aa = None
    counted = fmt == "counted"
    for lo in range(0, pairs, 32768):
        ch = L.reads(lo, min(32768, pairs - lo))
        if aa is None:
            f = 1.05 * pairs / ch.n_pairs
            aa = api.AllAlignments(loc, pairs, (int(ch.n_bases * f) + 2048) // 32 * 32, int(len(ch.recs) * f) + 4096,
                                   0 if counted else int(len(ch.cigar) * f) + 65536)
        aa.append(ch, counted=counted)
    aa.score(); ctx.synchronize()
    ctx.timing_reset()
    for _ in range(5):
        aa.score()
    ctx.synchronize()
    n, ms = ctx.timing(api.K_SCORE)
    st = aa.status()[0]
    sc = aa.run_filter()
    print(f"{fmt}: score kernel {ms / n:.2f} ms per launch ({n} launches); good pairs {int((st == cdefs.READ_GOOD).sum())}, "
          f"best genotype score {sc.max():.6f} at {int(np.argmax(sc))}", flush=True)


main()
