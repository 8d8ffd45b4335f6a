"""Candidate generation inside a locus (first slice; lcty_map.hip) against its Python restatement
(tests/pyref_map.py), against the generator's truth, and through the rest of the path."""
import numpy as np
import pytest

from locityper_amd import _lib, api, cdefs, synth
from tests import pyref_map as R

pytestmark = pytest.mark.gpu


def fastq_orientation(ch):
    """The generator's chunk holds SEQ as the BAM does (reverse-complemented for records on the reverse strand): back to the
    orientation the sequencer gave, and without records."""
    b2 = ch.bases2.copy(); nm = ch.nmask.copy()
    for pair in range(ch.n_pairs):
        j = int(ch.aln_off[pair])
        first = {0: None, 1: None}
        for i in range(int(ch.aln_off[pair]), int(ch.aln_off[pair + 1])):
            fl = int(ch.recs["flags"][i])
            e = 1 if fl & cdefs.FLAG_MATE2 else 0
            if first[e] is None and not fl & (cdefs.FLAG_SECONDARY | cdefs.FLAG_SUPPL):
                first[e] = fl
        for e in (0, 1):
            m = 2 * pair + e
            ln, off = int(ch.mate_len[m]), int(ch.mate_off[m])
            if ln == 0 or first[e] is None or not first[e] & cdefs.FLAG_REVERSE:
                continue
            bases = [(int(ch.bases2[(off + i) >> 4]) >> (2 * ((off + i) & 15))) & 3 for i in range(ln)]
            isn = [(int(ch.nmask[(off + i) >> 5]) >> ((off + i) & 31)) & 1 for i in range(ln)]
            for w in range((ln + 15) // 16):
                b2[(off >> 4) + w] = 0
            for w in range((ln + 31) // 32):
                nm[(off >> 5) + w] = 0
            for i in range(ln):
                src = ln - 1 - i
                b2[(off + i) >> 4] |= np.uint32((3 - bases[src]) << (2 * ((off + i) & 15)))
                nm[(off + i) >> 5] |= np.uint32(isn[src] << ((off + i) & 31))
    z = np.zeros(ch.n_pairs + 1, dtype=np.uint64)
    return cdefs.ReadsChunk(ch.mate_len, ch.mate_off, b2, nm, z, np.zeros(0, dtype=cdefs.ALN_REC_DTYPE), z, np.zeros(0, dtype=np.uint32))


def setup(gpu_ctx, n_alleles, n_pairs, base_len, seed):
    L = synth.SynthLocus(n_alleles, n_pairs, seed=seed, base_len=base_len)
    p = api.resolve_params(api.default_params(), L.bg)
    loc = api.Locus(gpu_ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    return L, p, loc


@pytest.mark.parametrize("basis,over", [([0, 1, 2, 3, 4, 5], {}), ([4, 1], {"stride": 11, "min_score": 120}), ([2], {"k": 11, "stride": 3, "min_votes": 3})])
def test_mapper_equals_its_restatement(gpu_ctx, basis, over):
    L, p, loc = setup(gpu_ctx, 6, 160, 5000, seed=3)
    truth = L.reads(0, 160)
    fq = fastq_orientation(truth)
    mp = api.map_params(**over)
    api.build_map_index(loc, basis, k=mp.k)
    got = api.map_reads(loc, fq, mp)
    aln_off, recs, cig_off, cigar, strands = R.map_chunk(fq, L.seqs, L.seq_off, basis, mp)
    assert np.array_equal(got.aln_off, aln_off) and np.array_equal(got.cigar_off, cig_off)
    assert np.array_equal(got.cigar, cigar)
    want = np.array(recs, dtype=[("pos", "<u4"), ("contig", "<u2"), ("flags", "<u2"), ("n_cigar", "<u4"), ("cigar_rel", "<u4")])
    for f in ("pos", "contig", "flags", "n_cigar", "cigar_rel"):
        assert np.array_equal(got.recs[f], want[f]), f
    # SEQ comes back in BAM orientation: reverse-complemented where the primary record is on the reverse strand
    for m in range(2 * fq.n_pairs):
        bases, isn = R.mate_bases(fq, m)
        out, outn = R.mate_bases(got, m)
        if strands[m]:
            bases, isn = [3 - b for b in reversed(bases)], list(reversed(isn))
        assert out == bases and outn == isn


def test_mapper_finds_the_generators_alignments_and_the_path_calls_the_genotype(gpu_ctx):
    """All alleles as the basis: every read end gets a record at the place the generator drew it from (same allele, same
    position, same strand), and the mapped chunk through scoring + prefilter gives the true genotype — as the generator's own
    chunk does."""
    L, p, loc = setup(gpu_ctx, 6, 2500, 9000, seed=8)
    truth = L.reads(0, 2500)
    fq = fastq_orientation(truth)
    mp = api.map_params()
    api.build_map_index(loc, list(range(6)), k=mp.k)
    got = api.map_reads(loc, fq, mp)
    # back in BAM orientation: the read ends the generator aligned come back as the generator wrote them (a read end it left
    # unmapped, or drew from elsewhere, has no orientation to agree on)
    same = sum(R.mate_bases(got, m) == R.mate_bases(truth, m) for m in range(0, 2 * truth.n_pairs, 7))
    assert same >= 0.93 * len(range(0, 2 * truth.n_pairs, 7)), same
    hit = total = hit_exact = total_exact = 0
    for pair in range(truth.n_pairs):
        mine = {(int(r["contig"]), int(r["pos"]), int(r["flags"]) & (cdefs.FLAG_REVERSE | cdefs.FLAG_MATE2))
                for r in got.recs[int(got.aln_off[pair]):int(got.aln_off[pair + 1])] if not int(r["flags"]) & cdefs.FLAG_UNMAPPED}
        for r in truth.recs[int(truth.aln_off[pair]):int(truth.aln_off[pair + 1])]:
            fl = int(r["flags"])
            if fl & cdefs.FLAG_UNMAPPED or int(r["n_cigar"]) == 0:
                continue
            words = [int(w) for w in truth.cigar[int(truth.cigar_off[pair]) + int(r["cigar_rel"]):][:int(r["n_cigar"])]]
            if any((w & 15) in (1, 2, 4) for w in words):
                continue                                                         # clipped or gapped in the generator: not this slice
            found = (int(r["contig"]), int(r["pos"]), fl & (cdefs.FLAG_REVERSE | cdefs.FLAG_MATE2)) in mine
            total += 1; hit += found
            if len(words) == 1:                                                  # the read end as it is on the allele
                total_exact += 1; hit_exact += found
    # exact placements are found — except for read ends that lie inside one of the generator's microsatellites (a tenth of a 9-kb
    # synthetic locus): all their seeds are repetitive and do not vote, and they are exact at many shifts anyway; placements with
    # mismatches lose seeds and, below half the votes of the read end's best candidate, are left out (as the reference's mapper
    # does with -S 0.5): recovery brings those alleles back
    assert total_exact > 3000 and hit_exact >= 0.85 * total_exact, (hit_exact, total_exact)
    assert hit >= 0.75 * total, (hit, total)
    aa = api.AllAlignments.load(loc, got)
    ref = api.AllAlignments.load(loc, truth)
    gts = api.generate_genotypes(6, 2)
    assert tuple(gts[int(np.argmax(aa.run_filter()))]) == L.true_genotype == tuple(gts[int(np.argmax(ref.run_filter()))])


def test_mapper_misuse_fails_loudly(gpu_ctx):
    L, p, loc = setup(gpu_ctx, 4, 50, 4000, seed=5)
    fq = fastq_orientation(L.reads(0, 50))
    with pytest.raises(_lib.LocityperError):
        api.map_reads(loc, fq, api.map_params())                            # no index yet
    with pytest.raises(_lib.LocityperError):
        api.build_map_index(loc, [9], k=15)                                 # not an allele of the locus
    with pytest.raises(_lib.LocityperError):
        api.build_map_index(loc, list(range(4)), k=40)
    api.build_map_index(loc, [0, 1], k=15)
    with pytest.raises(_lib.LocityperError):
        api.map_reads(loc, fq, api.map_params(k=13))                        # the index was built for another k


def test_basis_mapping_plus_alignment_recovery_carries_the_path(gpu_ctx):
    """The reference's --basis flow without the external mapper: the read ends are mapped onto two basis alleles only, the other
    four are reached through the haplotype-to-haplotype alignments (lcty_recover_alignments) — and the prefilter over the
    recovered table finds the genotype the reads were drawn from, as it does with the generator's own records."""
    from tests import oracle_ffi as O
    L, p, loc = setup(gpu_ctx, 6, 2500, 9000, seed=8)
    H = O.HapAlns(6, transfer_fails=100, max_div=0.1)
    for q, r, words, nm, ln in L.hap_alns():
        H.add(q, r, words)
    H.sort()
    loc.set_hap_alns(H.entries, transfer_fails=100, max_div=0.1)
    fq = fastq_orientation(L.reads(0, 2500))
    mp = api.map_params()
    api.build_map_index(loc, [1, 4], k=mp.k)
    mapped = api.map_reads(loc, fq, mp)
    assert set(np.unique(mapped.recs["contig"][(mapped.recs["flags"] & cdefs.FLAG_UNMAPPED) == 0]).tolist()) == {1, 4}
    aa = api.AllAlignments.load(loc, mapped)
    before = int(aa.pair_alns()[0][-1])
    n_rec = aa.recover()
    after = int(aa.pair_alns()[0][-1])
    assert n_rec > 4000 and after > 2 * before                                  # the other alleles came in
    gts = api.generate_genotypes(6, 2)
    assert tuple(gts[int(np.argmax(aa.run_filter()))]) == L.true_genotype


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_mapper_on_random_inputs_equals_its_restatement(gpu_ctx, seed):
    """Random alleles a few SNVs and an indel apart, read ends of 20..250 bases from either strand with errors, bases that are not
    ACGT, read ends shorter than a seed, foreign read ends, single-end pairs; random parameters. Records, CIGAR words and SEQ
    orientation must equal the restatement's."""
    from tests.helpers import locus_arrays, make_bg
    rng = np.random.default_rng(100 + seed)
    base = rng.choice(list(b"ACGT"), size=2400).astype(np.uint8)
    haps = []
    for a in range(5):
        h = base.copy()
        for q in rng.integers(50, 2350, size=6):
            h[q] = ord("ACGT"[(("ACGT".index(chr(h[q]))) + 1 + int(rng.integers(0, 3))) % 4])
        h = h.tolist()
        cut = int(rng.integers(300, 2000))
        if a % 2:
            del h[cut:cut + int(rng.integers(1, 9))]
        else:
            h[cut:cut] = rng.choice(list(b"ACGT"), size=int(rng.integers(1, 9))).tolist()
        haps.append(bytearray(h))
    bg = make_bg()
    p = api.resolve_params(api.default_params(), bg)
    seqs, seq_off, cflat, cnt_off, _ = locus_arrays(haps, 25)
    loc = api.Locus(gpu_ctx, seqs, seq_off, cflat, cnt_off, 25, bg, p)
    comp = bytes.maketrans(b"ACGT", b"TGCA")

    def read_end():
        kind = int(rng.integers(0, 10))
        ln = int(rng.integers(20, 251)) if kind else int(rng.integers(1, 15))
        if kind == 1:
            s = bytes(rng.choice(list(b"ACGT"), size=ln).tolist())                       # foreign
        else:
            h = haps[int(rng.integers(0, 5))]
            at = int(rng.integers(0, len(h) - ln))
            s = bytearray(h[at:at + ln])
            for _ in range(int(rng.integers(0, 4))):
                q = int(rng.integers(0, ln)); s[q] = ord("ACGTN"[int(rng.integers(0, 5))])
            s = bytes(s)
        return (s.translate(comp)[::-1] if rng.integers(0, 2) else s).decode()
    pairs = [{"seq1": read_end(), "seq2": read_end() if rng.integers(0, 6) else "", "recs": []} for _ in range(120)]
    ch = cdefs.ReadsChunk.from_pairs(pairs)
    basis = sorted(rng.choice(5, size=int(rng.integers(1, 6)), replace=False).tolist())
    mp = api.map_params(k=int(rng.integers(9, 20)), stride=int(rng.integers(4, 9)), min_votes=int(rng.integers(1, 4)),
                        min_score=int(rng.integers(0, 120)), max_occ=int(rng.integers(0, 3)) * 7, band=int(rng.choice([0, 3, 16])),
                        gap_open=int(rng.integers(2, 14)), gap_extend=int(rng.integers(0, 3)))
    api.build_map_index(loc, basis, k=mp.k)
    got = api.map_reads(loc, ch, mp)
    aln_off, recs, cig_off, cigar, strands = R.map_chunk(ch, seqs, seq_off, basis, mp)
    assert np.array_equal(got.aln_off, aln_off) and np.array_equal(got.cigar_off, cig_off) and np.array_equal(got.cigar, cigar)
    want = np.array(recs, dtype=[("pos", "<u4"), ("contig", "<u2"), ("flags", "<u2"), ("n_cigar", "<u4"), ("cigar_rel", "<u4")])
    for f in ("pos", "contig", "flags", "n_cigar", "cigar_rel"):
        assert np.array_equal(got.recs[f], want[f]), f
    for m in range(2 * ch.n_pairs):
        bases, isn = R.mate_bases(ch, m)
        if strands[m]:
            bases, isn = [3 - b for b in reversed(bases)], list(reversed(isn))
        assert R.mate_bases(got, m) == (bases, isn)


@pytest.mark.parametrize("k", [8, 16, 23, 31])
def test_index_of_short_and_long_seeds_and_alleles_with_other_bases(gpu_ctx, k):
    """The k-mer index of the basis alleles is ordered by the library's own radix sort (lcty_map_index.hip): one pass per byte of the key a
    k-mer of this k can differ in plus the top byte — 3 passes at k = 8, 8 at k = 31, an even or odd number of buffer swaps — and windows
    with a base that is not ACGT (runs of N inside the alleles) sort behind every k-mer. The mapper on such an index must give the
    restatement's records."""
    from tests.helpers import locus_arrays, make_bg
    rng = np.random.default_rng(500 + k)
    base = rng.choice(list(b"ACGT"), size=3000).astype(np.uint8)
    haps = []
    for a in range(4):
        h = base.copy()
        for q in rng.integers(50, 2950, size=8):
            h[q] = ord("ACGT"[(("ACGT".index(chr(h[q]))) + 1 + int(rng.integers(0, 3))) % 4])
        for _ in range(3):                                                           # runs of N, and single ones
            at = int(rng.integers(100, 2800)); h[at:at + int(rng.integers(1, 40))] = ord("N")
        haps.append(bytearray(h.tolist()))
    bg = make_bg()
    p = api.resolve_params(api.default_params(), bg)
    seqs, seq_off, cflat, cnt_off, _ = locus_arrays(haps, 25)
    loc = api.Locus(gpu_ctx, seqs, seq_off, cflat, cnt_off, 25, bg, p)
    comp = bytes.maketrans(b"ACGT", b"TGCA")

    def read_end():
        h = haps[int(rng.integers(0, 4))]
        ln = int(rng.integers(60, 251))
        at = int(rng.integers(0, len(h) - ln))
        s = bytes(c if c != ord("N") else ord("A") for c in h[at:at + ln])
        return (s.translate(comp)[::-1] if rng.integers(0, 2) else s).decode()
    pairs = [{"seq1": read_end(), "seq2": read_end(), "recs": []} for _ in range(60)]
    ch = cdefs.ReadsChunk.from_pairs(pairs)
    basis = [0, 2, 3]
    mp = api.map_params(k=k, stride=5, min_votes=2, band=16)
    api.build_map_index(loc, basis, k=mp.k)
    got = api.map_reads(loc, ch, mp)
    aln_off, recs, cig_off, cigar, strands = R.map_chunk(ch, seqs, seq_off, basis, mp)
    assert np.array_equal(got.aln_off, aln_off) and np.array_equal(got.cigar_off, cig_off) and np.array_equal(got.cigar, cigar)
    want = np.array(recs, dtype=[("pos", "<u4"), ("contig", "<u2"), ("flags", "<u2"), ("n_cigar", "<u4"), ("cigar_rel", "<u4")])
    for f in ("pos", "contig", "flags", "n_cigar", "cigar_rel"):
        assert np.array_equal(got.recs[f], want[f]), f
    assert int((got.recs["flags"] & cdefs.FLAG_UNMAPPED == 0).sum()) > 60              # the reads do map


def test_mapper_aligns_clipped_candidates_with_gaps(gpu_ctx):
    """Read ends across a 4-base deletion / a 3-base insertion relative to the allele: end to end with one D / I run on the device
    as in the restatement (tests/test_pyref_map.py derives these records by hand); without a band they stay clipped."""
    from tests.helpers import locus_arrays, make_bg
    rng = np.random.default_rng(21)
    allele = bytes(rng.choice(list(b"ACGT"), size=2000).tolist())
    bg = make_bg()
    p = api.resolve_params(api.default_params(), bg)
    seqs, seq_off, cflat, cnt_off, _ = locus_arrays([bytearray(allele)], 25)
    loc = api.Locus(gpu_ctx, seqs, seq_off, cflat, cnt_off, 25, bg, p)
    r_del = allele[500:580] + allele[584:654]
    ins = b"TTG" if allele[1079:1082] != b"TTG" else b"CCA"
    r_ins = allele[1000:1080] + ins + allele[1080:1147]
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    ch = cdefs.ReadsChunk.from_pairs([{"seq1": r_del.decode(), "seq2": r_ins.translate(comp)[::-1].decode(), "recs": []}])
    api.build_map_index(loc, [0], k=15)
    for band in (16, 0):
        mp = api.map_params(band=band)
        got = api.map_reads(loc, ch, mp)
        aln_off, recs, cig_off, cigar, strands = R.map_chunk(ch, seqs, seq_off, [0], mp)
        assert np.array_equal(got.cigar, cigar) and [tuple(int(x) for x in r) for r in got.recs.tolist()] == [tuple(r) for r in recs]
        ops = [[int(w) & 15 for w in got.cigar[int(r["cigar_rel"]):int(r["cigar_rel"]) + int(r["n_cigar"])]] for r in got.recs]
        if band:
            assert ops == [[7, 2, 7], [7, 1, 7]] and int(got.recs["flags"][1]) & cdefs.FLAG_REVERSE
            assert [int(x) for x in got.recs["pos"]] == [500, 1000]
        else:
            assert all(4 in o and 1 not in o and 2 not in o for o in ops)


def test_mapped_records_straight_into_the_batch(gpu_ctx):
    """lcty_reads_map_append (records, CIGAR words and bases copied device to device) leaves the batch as lcty_reads_append of the
    mapped chunk does: the same status, pair alignments and matrix after scoring; chunk after chunk."""
    L, p, loc = setup(gpu_ctx, 6, 1200, 9000, seed=9)
    fq = fastq_orientation(L.reads(0, 1200))
    mp = api.map_params()
    api.build_map_index(loc, [0, 2, 5], k=mp.k)
    halves = [fq.slice(0, 500), fq.slice(500, 1200)]
    mapped = [api.map_reads(loc, h, mp) for h in halves]
    via_host = api.AllAlignments.load(loc, mapped)
    direct = api.AllAlignments(loc, 1200, sum(h.n_bases for h in halves), sum(len(m.recs) for m in mapped), sum(len(m.cigar) for m in mapped))
    for h in halves:
        api.map_append(direct, h, mp)
    direct.score()
    assert direct.n_good() == via_host.n_good() > 800
    for x, y in zip(direct.status(), via_host.status()):
        assert np.array_equal(x, y)
    o1, p1 = direct.pair_alns(); o2, p2 = via_host.pair_alns()
    assert np.array_equal(o1, o2) and np.array_equal(p1, p2)
    assert np.array_equal(direct.best_aln_matrix(), via_host.best_aln_matrix())
    tiny = api.AllAlignments(loc, 1200, fq.n_bases, 10, 10)                  # no room for the records: refused, nothing written
    with pytest.raises(_lib.LocityperError):
        api.map_append(tiny, fq, mp)
