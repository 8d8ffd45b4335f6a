"""Candidate generation inside a locus, the long route (second slice; lcty_map_long.hip: read ends of any length,
up to 256 basis alleles) against its Python restatement (tests/pyref_map_long.py), against the generator's truth, and through the
rest of the path (BASELINE.json configs[2] from bases alone: no external mapper)."""
import numpy as np
import pytest

from locityper_amd import _lib, api, cdefs, synth
from tests import pyref_map_long as R
from tests.pyref_map import mate_bases
from tests.test_gpu_map import fastq_orientation

pytestmark = pytest.mark.gpu

REC = [("pos", "<u4"), ("contig", "<u2"), ("flags", "<u2"), ("n_cigar", "<u4"), ("cigar_rel", "<u4")]
OP = "MIDNSHP=X"


def cigar_score(words, mp):
    """what the mapper's scoring gives a CIGAR (both read ends reached unless clipped)"""
    sc = 0
    for i, w in enumerate(words):
        n, c = w >> 4, OP[w & 15]
        sc += {"=": mp.match * n, "X": -mp.mismatch * n, "I": -(mp.gap_open + mp.gap_extend * (n - 1)), "D": -(mp.gap_open + mp.gap_extend * (n - 1)), "S": 0}[c]
    return sc + mp.end_bonus * ((OP[words[0] & 15] != "S") + (OP[words[-1] & 15] != "S"))


def assert_equals_restatement(got, ch, seqs, seq_off, basis, mp, paired):
    aln_off, recs, cig_off, cigar, strands = R.map_chunk_long(ch, seqs, seq_off, basis, mp, paired=paired)
    assert np.array_equal(got.aln_off, aln_off) and np.array_equal(got.cigar_off, cig_off)
    want = np.array(recs, dtype=REC)
    for f in ("pos", "contig", "flags", "n_cigar", "cigar_rel"):
        assert np.array_equal(got.recs[f], want[f]), f
    assert np.array_equal(got.cigar, cigar)
    for m in range(2 * ch.n_pairs):
        bases, isn = mate_bases(ch, m)
        if strands[m]:
            bases, isn = [3 - b for b in reversed(bases)], list(reversed(isn))
        assert mate_bases(got, m) == (bases, isn)
    return recs, cigar, cig_off


def random_locus(gpu_ctx, rng, n_alleles, length, paired):
    from tests.helpers import locus_arrays, make_bg
    base = rng.choice(list(b"ACGT"), size=length).astype(np.uint8)
    haps = []
    for a in range(n_alleles):
        h = base.copy()
        for q in rng.integers(50, length - 50, size=8):
            h[q] = ord("ACGT"[(("ACGT".index(chr(h[q]))) + 1 + int(rng.integers(0, 3))) % 4])
        h = h.tolist()
        for _ in range(2):                                                       # two indels of 1..60 bases per allele
            cut = int(rng.integers(200, len(h) - 200))
            ln = int(rng.integers(1, 61))
            if rng.integers(0, 2):
                del h[cut:cut + ln]
            else:
                h[cut:cut] = rng.choice(list(b"ACGT"), size=ln).tolist()
        haps.append(bytearray(h))
    bg = make_bg()
    p = api.resolve_params(api.default_params(), bg)
    seqs, seq_off, cflat, cnt_off, _ = locus_arrays(haps, 25)
    loc = api.Locus(gpu_ctx, seqs, seq_off, cflat, cnt_off, 25, bg, p)
    return haps, seqs, seq_off, loc


def noisy_read(rng, haps, lo, hi):
    """a stretch of an allele with substitutions, small indels and bases that are not ACGT; either strand"""
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    h = haps[int(rng.integers(0, len(haps)))]
    ln = int(rng.integers(lo, hi))
    at = int(rng.integers(0, len(h) - ln))
    s = bytearray(h[at:at + ln])
    for _ in range(int(rng.integers(0, 2 + ln // 40))):
        q = int(rng.integers(0, len(s)))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            del s[q:q + int(rng.integers(1, 5))]
        elif kind == 1:
            s[q:q] = rng.choice(list(b"ACGT"), size=int(rng.integers(1, 5))).tolist()
        else:
            s[q] = ord("ACGTN"[int(rng.integers(0, 5))])
    s = bytes(s)
    return (s.translate(comp)[::-1] if rng.integers(0, 2) else s).decode()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8])
def test_long_route_on_random_inputs_equals_its_restatement(gpu_ctx, seed):
    """Random alleles some SNVs and indels of up to 60 bases apart; read ends of 20..900 bases from either strand with substitutions,
    indels and bases that are not ACGT, read ends shorter than a seed, foreign ones, absent second ends; random parameters (narrow
    bands, short look-back, tight chain limits). Records, CIGAR words and SEQ orientation must equal the restatement's."""
    rng = np.random.default_rng(300 + seed)
    haps, seqs, seq_off, loc = random_locus(gpu_ctx, rng, 5, 2600, paired=True)

    def read_end():
        kind = int(rng.integers(0, 12))
        if kind == 0:
            return bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(1, 15))).tolist()).decode()
        if kind == 1:
            return bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(100, 400))).tolist()).decode()        # foreign
        return noisy_read(rng, haps, 20, 900)
    pairs = [{"seq1": read_end(), "seq2": read_end() if rng.integers(0, 4) == 0 else "", "recs": []} for _ in range(28)]
    ch = cdefs.ReadsChunk.from_pairs(pairs)
    basis = sorted(rng.choice(5, size=int(rng.integers(1, 6)), replace=False).tolist())
    mp = api.map_params(long_reads=True, route=cdefs.MAP_ROUTE_LONG, k=int(rng.integers(9, 20)), stride=int(rng.integers(3, 20)),
                        min_votes=int(rng.integers(1, 4)), min_score=int(rng.choice([-(1 << 31), 0, 60])), max_occ=int(rng.integers(0, 3)) * 7,
                        band=int(rng.choice([2, 5, 16]) if seed <= 4 else rng.choice([0, 1, 3, 11])), gap_open=int(rng.integers(2, 14)), gap_extend=int(rng.integers(0, 3)),
                        mismatch=int(rng.integers(1, 9)), end_bonus=int(rng.integers(0, 12)),
                        chain_back=int(rng.choice([1, 3, 16, 64])), chain_gap=int(rng.choice([60, 300, 2000])), chain_skew=int(rng.choice([0, 20, 60])))
    api.build_map_index(loc, basis, k=mp.k)
    got = api.map_reads(loc, ch, mp)
    assert_equals_restatement(got, ch, seqs, seq_off, basis, mp, paired=True)


def test_long_route_by_hand(gpu_ctx):
    """The placements tests/test_pyref_map_long.py derives by hand (a read across 300 bases an allele lacks, a 7-base deletion, an
    overhang, a foreign read), on the device: equal to the restatement, and the hand-made facts again."""
    from tests.helpers import locus_arrays, make_bg
    from tests.test_pyref_map_long import setup as hand_setup
    allele, other, seqs, seq_off = hand_setup()
    bg = make_bg()
    p = api.resolve_params(api.default_params(), bg)
    seqs2, seq_off2, cflat, cnt_off, _ = locus_arrays([bytearray(allele), bytearray(other)], 25)
    assert bytes(seqs2) == bytes(seqs)
    loc = api.Locus(gpu_ctx, seqs2, seq_off2, cflat, cnt_off, 25, bg, p)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    r1 = allele[400:1900]
    r2 = allele[2500:4000].translate(comp)[::-1]
    r3 = bytearray(allele[4200:5200]); del r3[500:507]; r3[100] = ord("A") if r3[100] != ord("A") else ord("C")
    r4 = b"ACGTTGCAACGGTTAGCATG" + allele[0:700]
    r5 = bytes(np.random.default_rng(5).choice(list(b"ACGT"), size=900).tolist())
    ch = cdefs.ReadsChunk.from_pairs([{"seq1": r.decode(), "seq2": "", "recs": []} for r in (r1, r2, bytes(r3), r4, r5)])
    mp = api.map_params(long_reads=True, stride=8, min_votes=2, min_score=0)
    api.build_map_index(loc, [0, 1], k=mp.k)
    got = api.map_reads(loc, ch, mp)                                          # auto: read ends > 256 bases take the long route
    assert_equals_restatement(got, ch, seqs2, seq_off2, [0, 1], mp, paired=False)
    w = lambda i: [int(x) for x in got.cigar[int(got.cigar_off[i // 2]) + int(got.recs["cigar_rel"][i]):][:int(got.recs["n_cigar"][i])]]
    assert [int(x) for x in got.aln_off] == [0, 2, 4, 6, 8, 9]
    assert int(got.recs["pos"][0]) == 400 and w(0) == [(1500 << 4) | 7]
    assert int(got.recs["flags"][3]) == cdefs.FLAG_REVERSE | cdefs.FLAG_SECONDARY and [OP[x & 15] for x in w(3)] == ["=", "I", "="] and w(3)[1] >> 4 == 300
    assert [OP[x & 15] for x in w(4)] == ["=", "X", "=", "D", "="]
    assert w(6) == [(20 << 4) | 4, (700 << 4) | 7]
    assert int(got.recs["flags"][8]) & cdefs.FLAG_UNMAPPED


def test_short_read_ends_on_more_than_32_basis_alleles(gpu_ctx):
    """40 basis alleles: more than the short route's votes can tell apart, so 150-base read ends take the long route (auto); equal to
    the restatement, and the mapped chunk gives the genotype the reads were drawn from."""
    L = synth.SynthLocus(40, 2000, seed=12, base_len=6000)
    p = api.resolve_params(api.default_params(), L.bg)
    loc = api.Locus(gpu_ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    truth = L.reads(0, 2000)
    fq = fastq_orientation(truth)
    mp = api.map_params(stride=7)                                              # the short reads' scores, the long route's chains
    basis = list(range(40))
    api.build_map_index(loc, basis, k=mp.k)
    with pytest.raises(_lib.LocityperError):
        api.map_reads(loc, fq, api.map_params(route=cdefs.MAP_ROUTE_SHORT))
    some = fq.slice(0, 24)
    assert_equals_restatement(api.map_reads(loc, some, mp), some, L.seqs, L.seq_off, basis, mp, paired=True)
    got = api.map_reads(loc, fq, mp)
    aa = api.AllAlignments.load(loc, got)
    ref = api.AllAlignments.load(loc, truth)
    gts = api.generate_genotypes(40, 2)
    mine, theirs = tuple(int(x) for x in gts[int(np.argmax(aa.run_filter()))]), tuple(int(x) for x in gts[int(np.argmax(ref.run_filter()))])
    assert mine == L.true_genotype == theirs, (mine, theirs, L.true_genotype)


def test_10kb_reads_from_bases_alone_call_the_genotype(gpu_ctx):
    """BASELINE.json configs[2] at test size without an external mapper: 10-kb ONT reads (3 % errors) as the sequencer gave them ->
    the long route on every allele -> scoring + prefilter = the genotype the reads were drawn from. The records agree with the
    generator's: same allele, same strand, the position within a few bases (an indel at the very start moves it), CIGARs that consume
    the read, and a score (the mapper's own scoring) no worse than that of the generator's CIGAR, give or take a few operations."""
    n_alleles, n_reads = 8, 96
    L = synth.SynthLocus(n_alleles, n_reads, seed=synth.SEED + 5, technology=cdefs.TECH_NANOPORE, read_len=10_000, base_len=40_000)
    p = api.resolve_params(api.default_params(), L.bg)
    loc = api.Locus(gpu_ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    truth = L.reads(0, n_reads)
    fq = fastq_orientation(truth)
    mp = api.map_params(long_reads=True)
    api.build_map_index(loc, list(range(n_alleles)), k=mp.k)
    got = api.map_reads(loc, fq, mp)
    found = close = total = 0
    for pair in range(n_reads):
        mine = {}
        for r in got.recs[int(got.aln_off[pair]):int(got.aln_off[pair + 1])]:
            if int(r["flags"]) & cdefs.FLAG_UNMAPPED:
                continue
            words = [int(x) for x in got.cigar[int(got.cigar_off[pair]) + int(r["cigar_rel"]):][:int(r["n_cigar"])]]
            assert sum(x >> 4 for x in words if OP[x & 15] in "=XIS") == int(got.mate_len[2 * pair])
            mine[(int(r["contig"]), int(r["flags"]) & cdefs.FLAG_REVERSE)] = (int(r["pos"]), words)
        for r in truth.recs[int(truth.aln_off[pair]):int(truth.aln_off[pair + 1])]:
            if int(r["flags"]) & cdefs.FLAG_UNMAPPED or int(r["n_cigar"]) == 0:
                continue
            total += 1
            key = (int(r["contig"]), int(r["flags"]) & cdefs.FLAG_REVERSE)
            if key not in mine:
                continue
            found += 1
            tw = [int(x) for x in truth.cigar[int(truth.cigar_off[pair]) + int(r["cigar_rel"]):][:int(r["n_cigar"])]]
            clip = lambda ws: sum(x >> 4 for x in ws if OP[x & 15] == "S")
            pos, words = mine[key]
            close += abs(pos - int(r["pos"])) <= 40 + clip(tw) and cigar_score(words, mp) >= cigar_score(tw, mp) - 100
    assert total >= n_reads * n_alleles * 0.9 and found >= 0.98 * total and close >= 0.95 * found, (total, found, close)
    aa = api.AllAlignments.load(loc, got)
    ref = api.AllAlignments.load(loc, truth)
    gts = api.generate_genotypes(n_alleles, 2)
    assert aa.n_good() >= 0.9 * ref.n_good()
    assert tuple(gts[int(np.argmax(aa.run_filter()))]) == L.true_genotype == tuple(gts[int(np.argmax(ref.run_filter()))])


def test_10kb_reads_on_a_basis_plus_recovery(gpu_ctx):
    """The reference's --basis flow for long reads without the external mapper: 10-kb reads mapped onto 3 of 12 alleles (long route),
    the other nine reached through the haplotype-to-haplotype alignments; the prefilter finds the genotype."""
    from tests import oracle_ffi as O
    n_alleles, n_reads = 12, 96
    L = synth.SynthLocus(n_alleles, n_reads, seed=synth.SEED + 6, technology=cdefs.TECH_NANOPORE, read_len=10_000, base_len=40_000)
    p = api.resolve_params(api.default_params(), L.bg)
    loc = api.Locus(gpu_ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    H = O.HapAlns(n_alleles, transfer_fails=100, max_div=0.1)
    for q, r, words, nm, ln in L.hap_alns():
        H.add(q, r, words)
    H.sort()
    loc.set_hap_alns(H.entries, transfer_fails=100, max_div=0.1)
    fq = fastq_orientation(L.reads(0, n_reads))
    mp = api.map_params(long_reads=True)
    api.build_map_index(loc, [1, 5, 9], k=mp.k)
    aa = api.AllAlignments(loc, n_reads, fq.n_bases, 4 * n_reads, 4 * n_reads * 4000)
    api.map_append(aa, fq, mp)
    aa.score()
    n_rec = aa.recover()
    assert n_rec >= 6 * aa.n_good()
    gts = api.generate_genotypes(n_alleles, 2)
    assert tuple(gts[int(np.argmax(aa.run_filter()))]) == L.true_genotype


def test_long_route_misuse_fails_loudly(gpu_ctx):
    L = synth.SynthLocus(4, 20, seed=5, base_len=4000)
    p = api.resolve_params(api.default_params(), L.bg)
    loc = api.Locus(gpu_ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    fq = fastq_orientation(L.reads(0, 20))
    api.build_map_index(loc, [0, 1], k=15)
    for bad in ({"chain_back": 0}, {"chain_back": 70}, {"chain_gap": 0}, {"chain_gap": 10000}, {"chain_skew": 2000}, {"route": 7}, {"band": 20}):
        with pytest.raises(_lib.LocityperError):
            api.map_reads(loc, fq, api.map_params(long_reads=True, route=bad.pop("route", cdefs.MAP_ROUTE_LONG), **bad))


def test_long_route_records_straight_into_the_batch(gpu_ctx):
    """lcty_reads_map_append on the long route (records, CIGAR words and bases copied device to device, chunk after chunk) leaves the batch as
    lcty_reads_append of the mapped chunks does: the same status, pair alignments and matrix after scoring."""
    n_alleles, n_reads = 4, 64
    L = synth.SynthLocus(n_alleles, n_reads, seed=synth.SEED + 8, technology=cdefs.TECH_NANOPORE, read_len=6_000, base_len=30_000)
    p = api.resolve_params(api.default_params(), L.bg)
    loc = api.Locus(gpu_ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    fq = synth.sequencer_orientation(L.reads(0, n_reads, primaries_only=True))
    mp = api.map_params(long_reads=True)
    api.build_map_index(loc, [0, 2, 3], k=mp.k)
    halves = [fq.slice(0, 24), fq.slice(24, n_reads)]
    mapped = [api.map_reads(loc, h, mp) for h in halves]
    assert all(int(m.recs["n_cigar"].max()) > 256 for m in mapped)            # long CIGARs: the records are not the short route's
    via_host = api.AllAlignments.load(loc, mapped)
    direct = api.AllAlignments(loc, n_reads, sum(h.n_bases for h in halves), sum(len(m.recs) for m in mapped), sum(len(m.cigar) for m in mapped))
    for h in halves:
        api.map_append(direct, h, mp)
    direct.score()
    assert direct.n_good() == via_host.n_good() > 0.8 * n_reads
    for x, y in zip(direct.status(), via_host.status()):
        assert np.array_equal(x, y)
    o1, p1 = direct.pair_alns(); o2, p2 = via_host.pair_alns()
    assert np.array_equal(o1, o2) and np.array_equal(p1, p2)
    assert np.array_equal(direct.best_aln_matrix(), via_host.best_aln_matrix())


@pytest.mark.parametrize("stride,skew", [(16, 500), (96, 500), (200, 40)])
def test_kilobase_reads_equal_the_restatement(gpu_ctx, stride, skew):
    """3-kb ONT reads (3 % errors) on three alleles against the restatement, record for record: with seeds 96 or 200 bases apart the pieces
    between anchors are hundreds of rows long — direction bytes through the scratch of the wavefront, the walk back over several blocks,
    the bases of a segment staged more than once — which the short random inputs above do not reach."""
    L = synth.SynthLocus(3, 10, seed=synth.SEED + 21, technology=cdefs.TECH_NANOPORE, read_len=3_000, base_len=12_000)
    p = api.resolve_params(api.default_params(), L.bg)
    loc = api.Locus(gpu_ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    fq = synth.sequencer_orientation(L.reads(0, 10, primaries_only=True))
    mp = api.map_params(long_reads=True, stride=stride, chain_skew=skew, min_votes=2)
    basis = [0, 1, 2]
    api.build_map_index(loc, basis, k=mp.k)
    got = api.map_reads(loc, fq, mp)
    recs, cigar, cig_off = assert_equals_restatement(got, fq, L.seqs, L.seq_off, basis, mp, paired=False)
    assert sum(1 for r in recs if not r[2] & cdefs.FLAG_UNMAPPED) >= 20 and max(r[3] for r in recs) > 100


def test_long_route_on_256_basis_alleles_equals_the_restatement(gpu_ctx):
    """The widest form of the long route — 256 basis alleles: 512 (allele, strand) groups per read end, the widest group tables and chain
    scratch — record for record against the restatement: eight 2-3-kb ONT reads (3 % errors) onto every allele of a 256-allele locus
    (the shape BASELINE.json configs[2] maps; the Python side takes a minute or two)."""
    L = synth.SynthLocus(256, 8, seed=synth.SEED + 23, technology=cdefs.TECH_NANOPORE, read_len=2_500, base_len=8_000)
    p = api.resolve_params(api.default_params(), L.bg)
    loc = api.Locus(gpu_ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    fq = synth.sequencer_orientation(L.reads(0, 8, primaries_only=True))
    assert int(fq.mate_len[::2].min()) >= 2000
    mp = api.map_params(long_reads=True)
    basis = list(range(256))
    api.build_map_index(loc, basis, k=mp.k)
    got = api.map_reads(loc, fq, mp)
    recs, cigar, cig_off = assert_equals_restatement(got, fq, L.seqs, L.seq_off, basis, mp, paired=False)
    mapped = [r for r in recs if not r[2] & cdefs.FLAG_UNMAPPED]
    assert len(mapped) >= 8 * 250 and len({r[1] for r in mapped}) == 256          # every allele reached, by (nearly) every read
