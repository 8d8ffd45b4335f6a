"""Developer measurement: lcty_locus_create at the neighbourhood of 10-kb reads (256 alleles of 50 kb), K3 in its direct and sliding form."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locityper_amd import api, synth, cdefs

L = synth.SynthLocus(256, 16, seed=5, technology=cdefs.TECH_NANOPORE, read_len=10_000)
p = api.resolve_params(api.default_params(), L.bg)
ctx = api.Context(0)
api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p); ctx.synchronize()
for form in (0, 1, 0, 1):
    ctx.set_knob("contig_info_slide", form)
    t0 = time.perf_counter()
    loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    ctx.synchronize()
    print(f"neighbourhood {L.bg.neighb}, 256 alleles: {'sliding' if form else 'direct'} form, locus set-up {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
