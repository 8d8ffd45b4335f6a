"""Known answers for the Python restatement of the candidate-generation slice (tests/pyref_map.py): hand-made placements whose
records can be written down without running anything — the checker of tests/test_gpu_map.py is itself checked here (CPU tier)."""
import numpy as np

from locityper_amd.cdefs import ReadsChunk
from tests import pyref_map as R


class P:
    k, stride, min_votes, max_occ, match, mismatch, end_bonus, min_score, band, gap_open, gap_extend = 15, 5, 2, 0, 2, 8, 10, 50, 16, 12, 1


def test_known_placements():
    rng = np.random.default_rng(11)
    allele = bytes(rng.choice(list(b"ACGT"), size=3000).tolist())
    other = bytearray(allele); other[1500] = ord("A") if other[1500] != ord("A") else ord("C")       # one SNV apart
    seqs = np.frombuffer(allele + bytes(other), dtype=np.uint8)
    seq_off = np.array([0, 3000, 6000], dtype=np.uint64)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    r1 = bytearray(allele[700:850]); r1[40] = ord("A") if r1[40] != ord("A") else ord("C")            # forward, one sequencing error
    r2 = allele[1430:1580].translate(comp)[::-1]                                                     # reverse strand, across the SNV
    r3 = b"ACGTTGCAAC" + allele[0:140]                                                               # ten bases hang over the start
    r4 = bytes(rng.choice(list(b"ACGT"), size=150).tolist())                                         # from nowhere
    ch = ReadsChunk.from_pairs([{"seq1": bytes(r1).decode(), "seq2": r2.decode(), "recs": []},
                                {"seq1": r3.decode(), "seq2": r4.decode(), "recs": []}])
    aln_off, recs, cig_off, cigar, strands = R.map_chunk(ch, seqs, seq_off, [0, 1], P)
    words = lambda r, pair: [int(w) for w in cigar[int(cig_off[pair]) + r[4]:][:r[3]]]
    op = lambda n, c: (n << 4) | "MIDNSHP=X".index(c)
    assert list(aln_off) == [0, 4, 7]
    # pair 0, read end 1: primary on allele 0 (ties go to the smaller allele), secondary on allele 1; 40=1X109= on both
    assert recs[0][:3] == (700, 0, 0) and recs[1][:3] == (700, 1, R.FLAG_SECONDARY)
    assert words(recs[0], 0) == [op(40, "="), op(1, "X"), op(109, "=")] == words(recs[1], 0)
    # read end 2: reverse strand; exact on allele 0, the SNV at allele position 1500 = offset 70 on allele 1
    assert recs[2][:3] == (1430, 0, R.FLAG_REVERSE | R.FLAG_MATE2) and words(recs[2], 0) == [op(150, "=")]
    assert recs[3][:3] == (1430, 1, R.FLAG_REVERSE | R.FLAG_MATE2 | R.FLAG_SECONDARY)
    assert words(recs[3], 0) == [op(70, "="), op(1, "X"), op(79, "=")]
    assert strands[:2] == [0, 1]
    # pair 1: the overhang is soft-clipped (no end bonus on that side), the foreign read end is unmapped
    assert recs[4][:3] == (0, 0, 0) and words(recs[4], 1) == [op(10, "S"), op(140, "=")]
    assert recs[5][:3] == (0, 1, R.FLAG_SECONDARY)
    assert recs[6][:4] == (0, 0, R.FLAG_UNMAPPED | R.FLAG_MATE2, 0)


def test_repetitive_seeds_do_not_vote():
    unit = b"AATAGT"
    allele = bytes(np.random.default_rng(3).choice(list(b"ACGT"), size=400).tolist()) + unit * 60 + \
        bytes(np.random.default_rng(4).choice(list(b"ACGT"), size=400).tolist())
    seqs = np.frombuffer(allele, dtype=np.uint8)
    seq_off = np.array([0, len(allele)], dtype=np.uint64)
    inside = (unit * 25).decode()                                       # every seed has 50+ places: unmapped
    edge = allele[330:480].decode()                                     # 70 unique bases anchor it
    ch = ReadsChunk.from_pairs([{"seq1": inside, "seq2": edge, "recs": []}])
    aln_off, recs, cig_off, cigar, strands = R.map_chunk(ch, seqs, seq_off, [0], P)
    assert recs[0][2] & R.FLAG_UNMAPPED
    assert recs[1][:3] == (330, 0, R.FLAG_MATE2) and int(cigar[recs[1][4]]) == (150 << 4) | 7


def test_known_gapped_placements():
    """A read end across a 4-base deletion and one with a 3-base insertion relative to the allele: without gaps they are clipped at
    the indel, with gaps they are end to end (scores by hand: 2 per match, 12 + extend per further gap base, 10 per end reached)."""
    rng = np.random.default_rng(21)
    allele = bytes(rng.choice(list(b"ACGT"), size=2000).tolist())
    seqs = np.frombuffer(allele, dtype=np.uint8)
    seq_off = np.array([0, 2000], dtype=np.uint64)
    r_del = allele[500:580] + allele[584:654]                                  # 150 bases, the allele has 4 more in the middle
    ins = b"TTG" if allele[1079:1082] != b"TTG" else b"CCA"
    r_ins = allele[1000:1080] + ins + allele[1080:1147]                        # 150 bases, 3 of them not on the allele
    ch = ReadsChunk.from_pairs([{"seq1": r_del.decode(), "seq2": r_ins.decode(), "recs": []}])
    aln_off, recs, cig_off, cigar, strands = R.map_chunk(ch, seqs, seq_off, [0], P)
    op = lambda n, c: (n << 4) | "MIDNSHP=X".index(c)
    w0 = [int(w) for w in cigar[recs[0][4]:recs[0][4] + recs[0][3]]]
    w1 = [int(w) for w in cigar[recs[1][4]:recs[1][4] + recs[1][3]]]
    assert recs[0][:2] == (500, 0) and recs[1][:2] == (1000, 0)
    # the gap may sit anywhere inside a repeat of its flanks; its length and the totals are fixed
    assert [w & 15 for w in w0] == [7, 2, 7] and (w0[1] >> 4) == 4 and (w0[0] >> 4) + (w0[2] >> 4) == 150
    assert [w & 15 for w in w1] == [7, 1, 7] and (w1[1] >> 4) == 3 and (w1[0] >> 4) + (w1[2] >> 4) == 147
    class NoGaps(P):
        band = 0
    _, recs0, _, cigar0, _ = R.map_chunk(ch, seqs, seq_off, [0], NoGaps)
    assert [int(w) & 15 for w in cigar0[recs0[0][4]:recs0[0][4] + recs0[0][3]]] in ([7, 4], [4, 7])      # clipped at the indel
