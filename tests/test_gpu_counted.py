"""lcty_reads_append_counted: the alignment table as SURVEY.md section 8(d) counts it — 16-byte entries whose operations the caller
has counted (Alignment::count_region_operations_fast + limited_clipping, seq/aln.rs:288-317) — must give exactly what the record
path gives: same statuses, k-mer counts, weights, likelihood matrix, pair alignments, prefilter scores and solver chains."""
import numpy as np
import pytest

from locityper_amd import _lib, api, cdefs, synth
from tests import oracle_ffi as O
from tests.helpers import make_bg, random_alleles, locus_arrays

pytestmark = pytest.mark.gpu


def _same(a, b):
    sa, sb = a.status(), b.status()
    for x, y in zip(sa, sb):
        assert np.array_equal(x, y)
    assert np.array_equal(a.best_aln_matrix(), b.best_aln_matrix())
    (oa, pa), (ob, pb) = a.pair_alns(), b.pair_alns()
    assert np.array_equal(oa, ob)
    # the arena order of pairs is decided by an atomic cursor, the entries of a pair are not
    assert pa.tobytes() == pb.tobytes() or all(pa[int(oa[r]):int(oa[r + 1])].tobytes() == pb[int(ob[r]):int(ob[r + 1])].tobytes() for r in range(len(oa) - 1))
    assert np.array_equal(a.run_filter(), b.run_filter())


@pytest.mark.parametrize("n_alleles,n_pairs,tech,rl", [(16, 4000, cdefs.TECH_ILLUMINA, 150), (300, 300, cdefs.TECH_ILLUMINA, 150),
                                                        (8, 400, cdefs.TECH_NANOPORE, 5000)])
def test_counted_batch_equals_record_batch(gpu_ctx, n_alleles, n_pairs, tech, rl):
    L = synth.SynthLocus(n_alleles, n_pairs, technology=tech, read_len=rl, base_len=20_000)
    p = api.resolve_params(api.default_params(), L.bg)
    loc = api.Locus(gpu_ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    ch = L.reads(0, n_pairs)
    raw = api.AllAlignments.load(loc, ch)
    half = n_pairs // 2
    cnt = api.AllAlignments.load(loc, [ch.slice(0, half), ch.slice(half, n_pairs)], counted=True)      # two appended chunks
    _same(raw, cnt)
    # and against the oracle, which reads the records
    ol = O.OracleLocus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    oa = ol.load(ch)
    st, w, unm, uk = cnt.status()
    assert np.array_equal(st, oa.status) and np.array_equal(uk, oa.uniq_kmers) and cnt.n_good() == oa.n_good
    assert np.abs(cnt.best_aln_matrix() - oa.best_aln_matrix()).max() < 1e-9
    # the solver stages sit on the same products
    gts = api.generate_genotypes(n_alleles, 2)[:6]
    seeds = api.chain_seeds(3, 6)
    g = api.default_solver(cdefs.SOLVER_GREEDY)
    assert np.array_equal(api.solve_stage(raw, gts, g, 1, seeds)[2], api.solve_stage(cnt, gts, g, 1, seeds)[2])


def test_counted_edge_cases_and_misuse(gpu_ctx):
    """Clipping at the allele ends is limited by the caller (aln.rs:288-296); secondaries, reverse strands, unmapped primaries;
    mixing the two record forms or recovering alignments on a counted batch is refused."""
    alleles = random_alleles(3, 1500, seed=5)
    bg = make_bg()
    p = api.resolve_params(api.default_params(), bg)
    seqs, seq_off, counts, cnt_off, _ = locus_arrays(alleles, 25)
    loc = api.Locus(gpu_ctx, seqs, seq_off, counts, cnt_off, 25, bg, p)
    M2, REV, SEC, UNM = cdefs.FLAG_MATE2, cdefs.FLAG_REVERSE, cdefs.FLAG_SECONDARY, cdefs.FLAG_UNMAPPED
    a0 = alleles[0].decode()
    pairs = [
        {"seq1": a0[300:450], "seq2": a0[600:750], "recs": [(0, 300, 0, "150="), (1, 300, SEC, "100=1X49="), (0, 600, M2 | REV, "150="), (2, 598, M2 | REV | SEC, "2S148=")]},
        {"seq1": a0[0:150], "seq2": a0[1350:1500], "recs": [(0, 0, 0, "5S145="), (0, 1350, M2 | REV, "140=10S")]},      # clips at both allele ends
        {"seq1": a0[100:250], "seq2": a0[400:550], "recs": [(0, 100, UNM, "150="), (0, 400, M2 | REV, "150=")]},        # unmapped first primary
        {"seq1": a0[200:350], "seq2": a0[500:650], "recs": [(0, 200, 0, "70=2I78="), (1, 200, SEC, "60=3D90="), (0, 500, M2 | REV, "150=")]},
    ]
    ch = cdefs.ReadsChunk.from_pairs(pairs)
    raw = api.AllAlignments.load(loc, ch)
    cnt = api.AllAlignments.load(loc, ch, counted=True)
    _same(raw, cnt)
    c = ch.counted(loc.allele_len)
    assert (c[4, 3] >> 16) == 0 and (c[5, 3] >> 16) == 10                # "5S" at position 0 is limited to 0 bases; "10S" with 10 bases of the allele left counts in full
    assert (c[3, 3] >> 16) == 2 and (c[0, 0] >> 28) == 0 and (c[2, 0] >> 28) == 1 and (c[1, 0] >> 28) == 2
    with pytest.raises(_lib.LocityperError) as e:                         # no mixing
        cnt.append(ch)
    assert e.value.code == cdefs.ERR_INVALID_INPUT
    with pytest.raises(_lib.LocityperError) as e:
        raw.append(ch, counted=True)
    assert e.value.code == cdefs.ERR_INVALID_INPUT


def test_lean_kernel_equals_the_general_kernel(gpu_ctx):
    """Counted batches go through the lean scoring kernel (at most one saved alignment per contig and read end), the pairs it leaves
    through its two-slot form (a second saved alignment per contig and read end) and what that leaves through the general one (lcty_score.hip). The same batch with the knob "score_lean" 0 — the general kernel on every
    pair — must give the same products bit for bit, also when many pairs are left to the general kernel (several secondaries of a
    read end on one contig), for single-end reads, and against the oracle."""
    from tests.test_gpu_parity import random_pairs
    rng = np.random.default_rng(41)
    alleles = random_alleles(9, 4000, seed=8)
    bg = make_bg()
    p = api.resolve_params(api.default_params(), bg)
    seqs, seq_off, counts, cnt_off, _ = locus_arrays(alleles, 25)
    loc = api.Locus(gpu_ctx, seqs, seq_off, counts, cnt_off, 25, bg, p)
    ol = O.OracleLocus(seqs, seq_off, counts, cnt_off, 25, bg, p)
    # (a counted alignment cannot stand for a record with an empty CIGAR: those pairs stay with the record form)
    pairs = [q for q in random_pairs(rng, alleles, 1500) if all(c != "" or (fl & cdefs.FLAG_UNMAPPED) for _, _, fl, c in q["recs"])]
    assert len(pairs) > 1000
    ch = cdefs.ReadsChunk.from_pairs(pairs)
    cases = [ch]
    L = synth.SynthLocus(40, 3000, seed=77, base_len=12_000)                 # the benchmark's shape: one record per allele and end
    p2 = api.resolve_params(api.default_params(), L.bg)
    loc2 = api.Locus(gpu_ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p2)
    try:
        for locus, chunk in ((loc, ch), (loc2, L.reads(0, 3000))):
            gpu_ctx.set_knob("score_lean", 1)
            lean = api.AllAlignments.load(locus, chunk, counted=True)
            gpu_ctx.set_knob("score_lean", 0)
            general = api.AllAlignments.load(locus, chunk, counted=True)
            _same(lean, general)
            lean.score()                                                    # still with the knob at 0: the same batch through the other kernel
            _same(lean, general)
            # the lean kernel's other form: pass 1 leaves the record's index and pass 3 scores the record again (what loci of more
            # than 310 alleles get) instead of keeping pass 1's products in LDS
            gpu_ctx.set_knob("score_lean", 1); gpu_ctx.set_knob("score_lean_keep", 0)
            again = api.AllAlignments.load(locus, chunk, counted=True)
            _same(again, general)
            gpu_ctx.set_knob("score_lean_keep", -1)
            # without the two-slot form in between (round 4's flow: the lean kernel, then the general one on everything it leaves)
            gpu_ctx.set_knob("score_lean_two", 0)
            again = api.AllAlignments.load(locus, chunk, counted=True)
            _same(again, general)
            gpu_ctx.set_knob("score_lean_two", -1)
    finally:
        gpu_ctx.set_knob("score_lean", -1); gpu_ctx.set_knob("score_lean_keep", -1); gpu_ctx.set_knob("score_lean_two", -1)
    oa = ol.load(ch)
    cnt = api.AllAlignments.load(loc, ch, counted=True)
    st, w, unm, uk = cnt.status()
    assert np.array_equal(st, oa.status) and np.array_equal(uk, oa.uniq_kmers)
    assert np.abs(cnt.best_aln_matrix() - oa.best_aln_matrix()).max() < 1e-9
    off, pa = cnt.pair_alns()
    assert np.array_equal(off, oa.pa_off) and np.array_equal(pa["mid1"], oa.pair_alns["mid1"]) and np.array_equal(pa["contig"], oa.pair_alns["contig"])
