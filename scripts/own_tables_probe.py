"""How far apart are solver chains when each side uses its OWN tables (oracle: statrs' Lanczos lgamma; product: libm / ocml)?
Statistics behind the bound of tests/test_gpu_parity.py::test_config1_whole_path_against_the_oracle_pipeline."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from locityper_amd import api, cdefs, synth
from tests import oracle_ffi as O

ctx = api.Context(0)
for seed_off in range(4):
    cfg = synth.CONFIGS[1]
    L = synth.SynthLocus(cfg["n_alleles"], cfg["n_pairs"], seed=synth.SEED + seed_off)
    p = api.resolve_params(api.default_params(), L.bg)
    loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    ol = O.OracleLocus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    ch = L.reads(0, cfg["n_pairs"])
    aa = api.AllAlignments.load(loc, ch); oa = ol.load(ch)
    gts = api.generate_genotypes(8, 2)
    for kind, attempts in ((cdefs.SOLVER_GREEDY, 4), (cdefs.SOLVER_ANNEAL, 4)):
        seeds = api.chain_seeds(100 + kind, len(gts) * attempts)
        m, v, l = api.solve_stage(aa, gts, api.default_solver(kind), attempts, seeds)
        m2, v2, l2 = O.solve_stage(ol, oa, gts, api.default_solver(kind), attempts, seeds)
        rel = np.abs(l - l2) / np.abs(l2)
        sd = np.sqrt(np.maximum(v, v2))
        print(f"locus {seed_off} kind {kind}: chains exact (1e-9 rel) {np.mean(rel <= 1e-9):.3f}, within 1e-6 {np.mean(rel <= 1e-6):.3f}, max rel {rel.max():.2e}, "
              f"max |dmean| {np.abs(m - m2).max():.3f}, max |dmean|/sd {np.nanmax(np.abs(m - m2) / np.maximum(sd, 1e-300)):.3f}, "
              f"best genotype same {int(np.argmax(m)) == int(np.argmax(m2))}", flush=True)
