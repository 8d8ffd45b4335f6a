"""Streaming batches (lcty_reads_create_streaming): records of one chunk on the device at a time, products of every chunk kept.
Everything downstream must equal the resident batch bit for bit (AllAlignments::load is a loop over reads). Needs an MI355X."""
import numpy as np
import pytest

from locityper_amd import _lib, api, cdefs, synth
from tests import oracle_ffi as O
from tests.helpers import compare_gpu_to_oracle
from tests.test_gpu_parity import both_loci

pytestmark = pytest.mark.gpu


def same_products(a, b):
    for x, y in zip(a.status(), b.status()):
        assert np.array_equal(x, y)
    assert np.array_equal(a.best_aln_matrix(), b.best_aln_matrix())
    (oa, pa), (ob, pb) = a.pair_alns(), b.pair_alns()
    assert np.array_equal(oa, ob) and np.array_equal(pa, pb)
    assert a.n_good() == b.n_good()


@pytest.mark.parametrize("n_alleles,tech,rl,n_pairs,sizes", [
    (24, cdefs.TECH_ILLUMINA, 150, 5000, (1700, 300, 1, 2999)),        # ragged chunks, a chunk of one pair
    (12, cdefs.TECH_NANOPORE, 4000, 600, (200, 200, 200)),
])
def test_streaming_equals_resident(gpu_ctx, n_alleles, tech, rl, n_pairs, sizes):
    L = synth.SynthLocus(n_alleles, n_pairs, technology=tech, read_len=rl, seed=55)
    loc, ol, p = both_loci(gpu_ctx, L)
    chunks, lo = [], 0
    for n in sizes:
        chunks.append(L.reads(lo, n)); lo += n
    assert lo == n_pairs
    resident = api.AllAlignments.load(loc, chunks)
    stream = api.AllAlignments.load_streaming(loc, chunks)
    same_products(stream, resident)
    if L.bg.is_paired:
        assert np.array_equal(stream.run_filter(), resident.run_filter())
    # and both equal the oracle's single load
    compare_gpu_to_oracle(stream, ol.load(L.reads(0, n_pairs)))
    # solver stages on top of the streamed products
    gts = api.generate_genotypes(n_alleles, 2)[:9]
    seeds = api.chain_seeds(3, 2 * len(gts))
    for kind in (cdefs.SOLVER_GREEDY, cdefs.SOLVER_ANNEAL):
        sv = api.default_solver(kind)
        if kind == cdefs.SOLVER_ANNEAL:
            sv.anneal_steps, sv.plato_size = 1500, 1000
        assert np.array_equal(api.solve_stage(stream, gts, sv, 2, seeds)[2], api.solve_stage(resident, gts, sv, 2, seeds)[2])
    o1, c1 = api.assignment_counts(stream, gts[0], api.default_solver(cdefs.SOLVER_GREEDY), 2, seeds[:2])
    o2, c2 = api.assignment_counts(resident, gts[0], api.default_solver(cdefs.SOLVER_GREEDY), 2, seeds[:2])
    assert np.array_equal(o1, o2) and np.array_equal(c1, c2)


def test_streaming_rescoring_appends_and_limits(gpu_ctx):
    L = synth.SynthLocus(10, 3000, seed=56)
    loc, ol, p = both_loci(gpu_ctx, L)
    chunks = [L.reads(0, 1000), L.reads(1000, 500), L.reads(1500, 500), L.reads(2000, 1000)]
    resident = api.AllAlignments.load(loc, chunks)
    cap = lambda f: int(1.2 * max(f(c) for c in chunks))        # two chunks of 500 pairs need not be smaller than one of 1000
    s = api.AllAlignments(loc, 3000, (cap(lambda c: c.n_bases) + 31) // 32 * 32, cap(lambda c: len(c.recs)), cap(lambda c: len(c.cigar)),
                          streaming_chunk_pairs=1000)
    s.append(chunks[0]); s.score(); s.score()                  # a chunk scored twice: the arena cursor goes back to its start
    s.append(chunks[1]); s.append(chunks[2])                   # two appends before a score share the chunk buffers
    with pytest.raises(_lib.LocityperError):
        s.append(L.reads(0, 1))                                # 1001 pairs would be resident
    s.score(); s.score()
    s.append(chunks[3]); s.score()
    same_products(s, resident)
    with pytest.raises(_lib.LocityperError):
        s.append(L.reads(0, 1))                                # the batch is full
    with pytest.raises(_lib.LocityperError) as e:
        s.recover()
    assert e.value.code in (cdefs.ERR_UNSUPPORTED, cdefs.ERR_INVALID_INPUT)
    # an arena that is too small fails loudly
    tiny = api.AllAlignments(loc, 3000, (cap(lambda c: c.n_bases) + 31) // 32 * 32, cap(lambda c: len(c.recs)), cap(lambda c: len(c.cigar)),
                             streaming_chunk_pairs=1000, cap_pair_alns=5000)
    tiny.append(chunks[0])
    with pytest.raises(_lib.LocityperError):
        tiny.score(); tiny.status()


def test_streaming_long_reads_beyond_one_chunk_of_memory(gpu_ctx):
    """configs[2] shape (10-kb ONT reads x 256 alleles, ~600 KB of CIGAR words per read) in small: 4096 reads through chunk
    buffers of 512 — the way 1 M reads (600 GB of records) go through 288 GB of HBM. Products: size-independent properties."""
    n, A, chunk = 4096, 256, 512
    L = synth.SynthLocus(A, n, technology=cdefs.TECH_NANOPORE, read_len=10_000, seed=57)
    loc, ol, p = both_loci(gpu_ctx, L)
    c0 = L.reads(0, chunk)
    est = lambda x: int(x * 1.15) + 4096
    s = api.AllAlignments(loc, n, est(c0.n_bases) // 32 * 32 + 32, est(len(c0.recs)), est(len(c0.cigar)), streaming_chunk_pairs=chunk)
    first = None
    for lo in range(0, n, chunk):
        ch = c0 if lo == 0 else L.reads(lo, chunk)
        s.append(ch); s.score()
        if lo == 0:
            first = api.AllAlignments.load(loc, ch)
    st, w, unm, uk = s.status()
    st0, w0, unm0, uk0 = first.status()
    assert np.array_equal(st[:chunk], st0) and np.array_equal(w[:chunk], w0) and np.array_equal(uk[:2 * chunk], uk0)
    M = s.best_aln_matrix()
    assert np.array_equal(M[:, :first.n_good()], first.best_aln_matrix())
    assert (st == cdefs.READ_GOOD).sum() > 0.8 * n
    gts = api.generate_genotypes(A, 2)
    sc = O.run_filter(M, gts)
    assert tuple(gts[int(np.argmax(sc))]) == L.true_genotype


def test_streaming_with_alignment_recovery_per_chunk(gpu_ctx):
    """Alignment recovery on a streaming batch sits between the two scoring passes of every chunk (it needs the records of the
    pairs it looks at, nothing of other pairs): products equal those of the resident batch recovered in one go."""
    from tests.test_oracle_transfer import make_haps, hap_alns_for
    from tests.helpers import make_bg, locus_arrays
    from locityper_amd.cdefs import ReadsChunk
    rng = np.random.default_rng(15)
    haps = make_haps(rng, 5, 2600)
    bg = make_bg()
    p = api.resolve_params(api.default_params(), bg)
    seqs, seq_off, cflat, cnt_off, _ = locus_arrays([bytearray(h) for h in haps], 25)
    loc = api.Locus(gpu_ctx, seqs, seq_off, cflat, cnt_off, 25, bg, p)
    H = hap_alns_for(haps, transfer_fails=3)
    loc.set_hap_alns(H.entries, transfer_fails=3, max_div=0.2)
    M2, REV = cdefs.FLAG_MATE2, cdefs.FLAG_REVERSE
    pairs = []
    for _ in range(300):
        src = int(rng.integers(0, 5))
        p1 = int(rng.integers(320, len(haps[src]) - 800)); p2 = p1 + int(rng.integers(200, 420))
        pairs.append({"seq1": haps[src][p1:p1 + 150].decode(), "seq2": haps[src][p2:p2 + 150].decode(),
                      "recs": [(src, p1, 0, "150="), (src, p2, M2 | REV, "150=")]})
    resident = api.AllAlignments.load(loc, ReadsChunk.from_pairs(pairs))
    n_all = resident.recover()
    assert n_all > 1500
    chunks = [ReadsChunk.from_pairs(pairs[a:b]) for a, b in ((0, 120), (120, 121), (121, 300))]
    cap = lambda f: int(1.1 * max(f(c) for c in chunks)) + 64
    s = api.AllAlignments(loc, 300, (cap(lambda c: c.n_bases) + 31) // 32 * 32, cap(lambda c: len(c.recs)), cap(lambda c: len(c.cigar)),
                          streaming_chunk_pairs=179, cap_pair_alns=300 * 5 * 10)
    n_stream = 0
    for c in chunks:
        s.append(c); s.score()
        n_stream += s.recover()                                  # transfers + the second scoring pass of the chunk
    assert n_stream == n_all
    same_products(s, resident)
    assert np.array_equal(s.run_filter(), resident.run_filter())
    # against the oracle's load with recovery
    ol = O.OracleLocus(seqs, seq_off, cflat, cnt_off, 25, bg, p)
    compare_gpu_to_oracle(s, ol.load_recover(ReadsChunk.from_pairs(pairs), H), index_fields=())
