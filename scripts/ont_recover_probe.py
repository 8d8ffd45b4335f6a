"""Developer measurement: repeated lcty_recover_alignments on a batch of synthetic 10-kb ONT reads (wall vs kernel time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locityper_amd import api, synth, cdefs

def main():
    nont = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
    A = 256
    ctx = api.Context(0)
    Lo = synth.SynthLocus(A, nont, seed=synth.SEED + 77, technology=cdefs.TECH_NANOPORE, read_len=10_000)
    po = api.resolve_params(api.default_params(), Lo.bg)
    loco = api.Locus(ctx, Lo.seqs, Lo.seq_off, Lo.counts, Lo.cnt_off, Lo.k, Lo.bg, po)
    loco.set_hap_alns(Lo.hap_alns(), transfer_fails=100, max_div=0.1)
    prim = [Lo.reads(lo, min(256, nont - lo), primaries_only=True) for lo in range(0, nont, 256)]
    for rep in range(4):
        ao = api.AllAlignments.load(loco, prim)
        ctx.timing_reset()
        t = time.perf_counter()
        n = ao.recover()
        dt = time.perf_counter() - t
        print(f"call {rep}: wall {dt:.3f} s, transfer kernel {ctx.timing(api.K_TRANSFER)[1]:.1f} ms x{ctx.timing(api.K_TRANSFER)[0]}, score {ctx.timing(api.K_SCORE)[1]:.1f} ms, new {n}", flush=True)
        ao.close()

main()
