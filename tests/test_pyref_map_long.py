"""Known answers for the Python restatement of the long route of candidate generation (tests/pyref_map_long.py): placements whose
records can be written down by hand — the checker of tests/test_gpu_map_long.py is itself checked here (CPU tier)."""
import numpy as np

from locityper_amd.cdefs import ReadsChunk
from tests import pyref_map_long as R


class P:
    k, stride, min_votes, max_occ, match, mismatch, end_bonus, min_score, band, gap_open, gap_extend = 15, 8, 2, 0, 2, 4, 10, 0, 16, 6, 2
    chain_gap, chain_skew, chain_back = 2000, 500, 32


OP = "MIDNSHP=X"
op = lambda n, c: (n << 4) | OP.index(c)


def words_of(rec, pair, cig_off, cigar):
    return [int(w) for w in cigar[int(cig_off[pair]) + rec[4]:][:rec[3]]]


def consumed(words):
    q = sum(w >> 4 for w in words if OP[w & 15] in "=XIS")
    t = sum(w >> 4 for w in words if OP[w & 15] in "=XD")
    return q, t


def setup():
    rng = np.random.default_rng(17)
    allele = bytes(rng.choice(list(b"ACGT"), size=6000).tolist())
    other = bytearray(allele)
    del other[3000:3300]                                                         # the second allele lacks 300 bases
    other[1000] = ord("A") if other[1000] != ord("A") else ord("C")
    seqs = np.frombuffer(allele + bytes(other), dtype=np.uint8)
    seq_off = np.array([0, 6000, 6000 + len(other)], dtype=np.uint64)
    return allele, bytes(other), seqs, seq_off


def test_long_read_ends_by_hand():
    allele, other, seqs, seq_off = setup()
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    r1 = allele[400:1900]                                                        # 1 500 bases as they are on allele 0; one X on allele 1
    r2 = allele[2500:4000].translate(comp)[::-1]                                 # reverse strand, across the 300 bases allele 1 lacks
    r3 = bytearray(allele[4200:5200]); del r3[500:507]; r3[100] = ord("A") if r3[100] != ord("A") else ord("C")
    r4 = b"ACGTTGCAACGGTTAGCATG" + allele[0:700]                                 # twenty bases hang over the start
    r5 = bytes(np.random.default_rng(5).choice(list(b"ACGT"), size=900).tolist())  # from nowhere
    ch = ReadsChunk.from_pairs([{"seq1": r.decode(), "seq2": "", "recs": []} for r in (r1, r2, bytes(r3), r4, r5)])
    aln_off, recs, cig_off, cigar, strands = R.map_chunk_long(ch, seqs, seq_off, [0, 1], P, paired=False)
    assert list(aln_off) == [0, 2, 4, 6, 8, 9]
    # r1: end to end on both alleles
    assert recs[0][:3] == (400, 0, 0) and words_of(recs[0], 0, cig_off, cigar) == [op(1500, "=")]
    assert recs[1][:3] == (400, 1, R.FLAG_SECONDARY) and words_of(recs[1], 0, cig_off, cigar) == [op(600, "="), op(1, "X"), op(899, "=")]
    # r2: exact on allele 0 (reverse strand); on allele 1 the 300 bases are an insertion of the read (somewhere inside the repeat of its flanks)
    assert recs[2][:3] == (2500, 0, R.FLAG_REVERSE) and words_of(recs[2], 1, cig_off, cigar) == [op(1500, "=")]
    w = words_of(recs[3], 1, cig_off, cigar)
    assert recs[3][:3] == (2500, 1, R.FLAG_REVERSE | R.FLAG_SECONDARY)
    assert [OP[x & 15] for x in w] == ["=", "I", "="] and w[1] >> 4 == 300 and consumed(w) == (1500, 1200)
    assert [strands[0], strands[2]] == [0, 1]                                    # per read end, the absent second ends included
    # r3: a mismatch and a 7-base deletion relative to allele 0
    w = words_of(recs[4], 2, cig_off, cigar)
    assert recs[4][:3] == (4200, 0, 0)
    assert [OP[x & 15] for x in w] == ["=", "X", "=", "D", "="] and w[0] >> 4 == 100 and w[3] >> 4 == 7 and consumed(w) == (993, 1000)
    assert recs[5][:3] == (3900, 1, R.FLAG_SECONDARY)
    # r4: the overhang is clipped
    assert recs[6][:3] == (0, 0, 0) and words_of(recs[6], 3, cig_off, cigar) == [op(20, "S"), op(700, "=")]
    # r5: unmapped
    assert recs[8][2] & R.FLAG_UNMAPPED and recs[8][3] == 0


def test_scores_add_up():
    """The score of a candidate is the sum over its CIGAR: 2 per =, -4 per X, a gap of n bases 6 + 2 (n - 1), 10 per read end reached."""
    allele, other, seqs, seq_off = setup()
    index = R.build_index(seqs, seq_off, [1], P.k)
    r = bytearray(allele[2200:3800])                                             # on allele 1: 300 bases of the read are an insertion
    r[50] = ord("A") if r[50] != ord("A") else ord("C")
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    bases = [code[c] for c in r]
    found = R.map_mate_long(bases, [False] * len(r), index, seqs, seq_off, [1], P)
    assert len(found) == 1
    allele_ix, strand, pos, score, cig = found[0]
    assert (allele_ix, strand, pos) == (1, 0, 2200)
    want = 0
    for w in cig:
        n, c = w >> 4, OP[w & 15]
        want += {"=": 2 * n, "X": -4 * n, "I": -(6 + 2 * (n - 1)), "D": -(6 + 2 * (n - 1)), "S": 0}[c]
    assert score == want + 20 and consumed(cig) == (1600, 1300)


def test_sequencer_orientation_of_the_generator_equals_the_plain_loop():
    """locityper_amd.synth.sequencer_orientation (numpy, used by bench.py and the probes) against the base-by-base loop the GPU tests use."""
    from locityper_amd import cdefs, synth
    from tests.test_gpu_map import fastq_orientation
    for kw in ({"n_alleles": 5, "n_pairs": 120, "seed": 3, "base_len": 5000},
               {"n_alleles": 3, "n_pairs": 12, "seed": 4, "technology": cdefs.TECH_NANOPORE, "read_len": 6000, "base_len": 20000}):
        L = synth.SynthLocus(**kw)
        ch = L.reads(0, kw["n_pairs"], primaries_only="technology" in kw)
        a, b = fastq_orientation(ch), synth.sequencer_orientation(ch)
        assert np.array_equal(a.bases2, b.bases2) and np.array_equal(a.nmask, b.nmask) and len(b.recs) == 0
        assert not np.array_equal(ch.bases2, b.bases2)                                # some read ends were on the reverse strand
