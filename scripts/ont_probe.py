"""Developer measurement: BASELINE configs[2]-shaped sample (single-end 10 kb ONT reads x 256 alleles) through the scoring
kernel and the prefilter (parity of this shape is tests/test_gpu_parity.py's business)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from locityper_amd import api, synth, cdefs

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    A = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    t0 = time.time()
    L = synth.SynthLocus(A, n, technology=cdefs.TECH_NANOPORE, read_len=10_000)
    p = api.resolve_params(api.default_params(), L.bg)
    ctx = api.Context(0)
    loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    chunks = [L.reads(lo, min(256, n - lo)) for lo in range(0, n, 256)]
    aa = api.AllAlignments(loc, n, sum(c.n_bases for c in chunks) + 64, sum(len(c.recs) for c in chunks), sum(len(c.cigar) for c in chunks))
    for c in chunks: aa.append(c)
    print(f"generated {n} reads x {A} alleles in {time.time()-t0:.1f} s: {sum(len(c.recs) for c in chunks)} records, "
          f"{sum(len(c.cigar) for c in chunks)/1e6:.1f} M CIGAR words", flush=True)
    aa.score(); ctx.synchronize(); ctx.timing_reset()
    for _ in range(3): aa.score()
    k, ms = ctx.timing(api.K_SCORE)
    sc = aa.run_filter()
    gts = api.generate_genotypes(A, 2)
    bytes_in = 4 * sum(len(c.cigar) for c in chunks) + 16 * sum(len(c.recs) for c in chunks)
    print(f"score_reads {ms/k:.3f} ms -> {n/(ms/k)*1e3:.0f} reads/s, {bytes_in/(ms/k)/1e6:.0f} GB/s of records+CIGAR; good {aa.n_good()}; "
          f"call {tuple(gts[int(np.argmax(sc))])} true {L.true_genotype}", flush=True)

main()
