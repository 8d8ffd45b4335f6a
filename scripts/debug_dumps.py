#!/usr/bin/env python3
"""The reference's `--debug 2` files of one locus, written from the CPU oracle and from the HIP path:

    read_kmers.csv   read_hash uniq_kmers1 uniq_kmers2 weight        UniqueKmers::calculate_read_weight (model/locs.rs:975-999; header 1062-1063)
    read_pairs.csv   read_hash contig pos1 pos2 lik                  GrouppedAlignments::write_read_pair_info (locs.rs:648-665, 1215-1231)
    sol.csv          stage genotype score                            run_filter rows "0" (solvers/solve.rs:116), stage rows (841; header 937-938)

These are the files a `locityper genotype --debug 2` run leaves in OUT/loci/<locus>/ (brotli-compressed there), so they are the hook
by which this restatement can be held against a Rust run on the same inputs if a toolchain ever becomes available (SURVEY.md
section 4 / 8d). Differences that are known and deliberate: read_hash is the index of the read pair in the input (the reference
hashes the read name with wyhash, unpinned here), contig names are a0, a1, ..; read_kmers.csv has the rows of the reads that
end up used or with few k-mers (a read rejected after calculate_read_weight leaves no trace in either library's outputs); the
solver rows depend on the injected random-number definitions (oracle/lcty_oracle.h).

Usage: python scripts/debug_dumps.py OUTDIR [--alleles 8 --pairs 10000]   (needs a GPU; writes OUTDIR/{oracle,hip}/*.csv and compares)
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

INV_LN10 = 1.0 / math.log(10.0)
READ_GOOD, READ_FEW_KMERS = 0, 3


def _codes():
    from locityper_amd import cdefs
    return cdefs.READ_GOOD, cdefs.READ_FEW_KMERS


def read_kmers_csv(status, uniq_kmers, params, paired):
    """calculate_read_weight rows (locs.rs:975-999): k-mer counts of the mates and the clamped k-mer weight."""
    good, few = _codes()
    mult = 1.0 / (params.kmer_soft_thresh + 1.0 - params.kmer_hard_thresh)           # UniqueKmers::new, locs.rs:956-958
    interc = (1.0 - params.kmer_hard_thresh) * mult
    out = ["read_hash\tuniq_kmers1\tuniq_kmers2\tweight"]
    for r in np.flatnonzero((status == good) | (status == few)):
        c1, c2 = int(uniq_kmers[2 * r]), int(uniq_kmers[2 * r + 1])
        w = min(max(interc + float((c1 + (c2 if paired else 0)) & 0xFFFF) * mult, 0.0), 1.0)
        out.append(f"{r}\t{c1}\t{c2 if paired else '*'}\t{w:.2f}")
    return "\n".join(out) + "\n"


def read_pairs_csv(status, unmapped_prob, pa_off, pair_alns, rec_pos, aln_off):
    """write_read_pair_info::<false> (locs.rs:1215-1231): the PairAlignments of the used reads, then their unmapped row."""
    good, _ = _codes()
    out = ["read_hash\tcontig\tpos1\tpos2\tlik"]
    for r in np.flatnonzero(status == good):
        base = int(aln_off[r])
        for t in range(int(pa_off[r]), int(pa_off[r + 1])):
            pa = pair_alns[t]
            p1 = "*" if int(pa["ix1"]) == 0xFFFFFFFF else str(int(rec_pos[base + int(pa["ix1"])]) + 1)
            p2 = "*" if int(pa["ix2"]) == 0xFFFFFFFF else str(int(rec_pos[base + int(pa["ix2"])]) + 1)
            out.append(f"{r}\ta{int(pa['contig'])}\t{p1}\t{p2}\t{float(pa['ln_prob']) * INV_LN10:.4f}")
        out.append(f"{r}\t*\t*\t*\t{float(unmapped_prob[r]) * INV_LN10:.4f}")
    return "\n".join(out) + "\n"


def sol_csv(gts, filter_scores, stage_rows):
    """sol.csv: "0" rows of run_filter for every genotype (solve.rs:116: {:.3}), then (stage, genotype, mean) rows ({:.4}, 841)."""
    name = lambda g: ",".join(f"a{int(a)}" for a in g)
    out = ["stage\tgenotype\tscore"]
    if filter_scores is not None:
        out += [f"0\t{name(g)}\t{s * INV_LN10:.3f}" for g, s in zip(gts, filter_scores)]
    for stage, g, mean in stage_rows:
        out.append(f"{stage}\t{name(g)}\t{mean * INV_LN10:.4f}")
    return "\n".join(out) + "\n"


def same_but_last_digit(a, b, tol):
    """Two dumps agree: same lines, same text fields, numbers within `tol` (a value on a rounding boundary may print differently)."""
    la, lb = a.splitlines(), b.splitlines()
    if len(la) != len(lb):
        return False, f"{len(la)} vs {len(lb)} lines"
    for i, (x, y) in enumerate(zip(la, lb)):
        if x == y:
            continue
        fx, fy = x.split("\t"), y.split("\t")
        if len(fx) != len(fy) or fx[:-1] != fy[:-1] or abs(float(fx[-1]) - float(fy[-1])) > tol:
            return False, f"line {i}: {x!r} vs {y!r}"
    return True, ""


def dumps_of_both(ctx, n_alleles, n_pairs, scheme=(("greedy", 12, 2), ("anneal", 4, 4)), base_len=50_000):
    """{'oracle': {file: text}, 'hip': {file: text}} for one synthetic locus; the solver rows of both sides use the device's tables
    (tests/test_gpu_solve.py: the chains then follow each other move for move)."""
    from locityper_amd import api, cdefs, synth
    from tests import oracle_ffi as O
    L = synth.SynthLocus(n_alleles, n_pairs, base_len=base_len)
    p = api.resolve_params(api.default_params(), L.bg)
    ch = L.reads(0, n_pairs)
    loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    aa = api.AllAlignments.load(loc, ch)
    ol = O.OracleLocus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    oa = ol.load(ch)
    gts = api.generate_genotypes(n_alleles, 2)
    paired = bool(L.bg.is_paired)
    res = {}
    # HIP side
    st, w, unm, uk = aa.status()
    off, pa = aa.pair_alns()
    sc = aa.run_filter()
    sides = {"hip": (st, unm, uk, off, pa, sc), "oracle": (oa.status, oa.unmapped_prob, oa.uniq_kmers, oa.pa_off, oa.pair_alns,
                                                           O.run_filter(oa.best_aln_matrix(), gts))}
    ol.inject_tables(loc.depth_lut(), loc.window_weights())
    oa_inj = O.alns_from_arrays(n_alleles, st, w, unm, off, pa)
    for side, (s_, unm_, uk_, off_, pa_, sc_) in sides.items():
        rows = []
        keep = np.argsort(-sc_, kind="stable")[:scheme[0][1]]
        for si, (kind, in_size, attempts) in enumerate(scheme):
            keep = keep[:in_size]
            solver = api.default_solver(cdefs.SOLVER_GREEDY if kind == "greedy" else cdefs.SOLVER_ANNEAL)
            seeds = api.chain_seeds(500 + si, len(keep) * attempts)
            if side == "hip":
                m, _, _ = api.solve_stage(aa, gts[keep], solver, attempts, seeds)
            else:
                m, _, _ = O.solve_stage(ol, oa_inj, gts[keep], solver, attempts, seeds)
            rows += [(si + 1, gts[g], float(x)) for g, x in zip(keep, m)]
            keep = keep[np.argsort(-m, kind="stable")]
        res[side] = {"read_kmers.csv": read_kmers_csv(s_, uk_, p, paired),
                     "read_pairs.csv": read_pairs_csv(s_, unm_, off_, pa_, ch.recs["pos"], ch.aln_off),
                     "sol.csv": sol_csv(gts, sc_, rows)}
    return res


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("--alleles", type=int, default=8)
    ap.add_argument("--pairs", type=int, default=10_000)
    a = ap.parse_args()
    from locityper_amd import api
    ctx = api.Context(0)
    res = dumps_of_both(ctx, a.alleles, a.pairs)
    for side, files in res.items():
        os.makedirs(os.path.join(a.outdir, side), exist_ok=True)
        for name, text in files.items():
            open(os.path.join(a.outdir, side, name), "w").write(text)
            if side == "hip":
                # ... and as the reference leaves them in OUT/loci/<locus>/: brotli streams (`sol.csv.br`, `read_pairs.csv.br`, ...:
                # solvers/solve.rs:937-938, model/locs.rs:1062-1065) through the library's writer
                from locityper_amd import io as lio
                lio.write_br(os.path.join(a.outdir, side, name + ".br"), text.encode())
    for name in res["hip"]:
        ok, why = same_but_last_digit(res["hip"][name], res["oracle"][name], 2e-4 if name != "read_kmers.csv" else 0.011)
        print(f"{name}: {'same' if ok else 'DIFFERENT ' + why} ({res['hip'][name].count(chr(10))} lines)")


if __name__ == "__main__":
    main()
