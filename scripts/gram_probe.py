"""Developer measurement: the prefilter at the shard shape of BASELINE configs[4] (131 072 read pairs x 4 096 alleles) — the f64 tile
kernel against the integer Gram contraction on the matrix cores (lcty_gram.hip). usage: gram_probe.py [pairs] [alleles]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from locityper_amd import api, synth


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    A = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    L = synth.SynthLocus(A, pairs, seed=synth.SEED, base_len=3000 if A >= 2048 else 6000)
    p = api.resolve_params(api.default_params(), L.bg)
    ctx = api.Context(0)
    loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    aa = None
    chunk = 4096
    for lo in range(0, pairs, chunk):
        ch = L.reads(lo, min(chunk, pairs - lo))
        if aa is None:
            f = 1.1 * pairs / ch.n_pairs
            aa = api.AllAlignments(loc, pairs, (int(ch.n_bases * f) + 2048) // 32 * 32, int(len(ch.recs) * f) + 4096, int(len(ch.cigar) * f) + 65536)
        aa.append(ch)
    aa.score()
    out = {"workload": f"{pairs} read pairs x {A} alleles (G = {A * (A + 1) // 2})"}
    res = {}
    for name, knob in (("f64_tile", 0), ("gram_mfma", 1)):
        ctx.set_knob("prefilter_gram", knob)
        aa.prefilter_async(); ctx.synchronize()                           # warm-up: buffers
        ctx.timing_reset()
        t = time.perf_counter()
        for _ in range(3):
            aa.prefilter_async()
        ctx.synchronize()
        wall = (time.perf_counter() - t) / 3
        res[name] = aa.prefilter_scores()
        out[name] = {"wall_ms": wall * 1e3, "kernel_ms": ctx.timing(api.K_PREFILTER)[1] / 3}
    d = np.abs(res["gram_mfma"] - res["f64_tile"]).max() / np.abs(res["f64_tile"]).max()
    out["max_rel_diff"] = float(d)
    out["same_best"] = bool(np.argmax(res["gram_mfma"]) == np.argmax(res["f64_tile"]))
    print(json.dumps(out), flush=True)


main()
