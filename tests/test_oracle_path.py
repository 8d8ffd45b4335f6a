"""Oracle (C) known-answer tests + cross-check against the independent Python transliteration
(tests/pyref.py). CPU only."""
import math

import numpy as np
import pytest

from locityper_amd import cdefs, synth
from locityper_amd.cdefs import ReadsChunk
from tests import oracle_ffi as O
from tests import pyref
from tests.helpers import make_bg, random_alleles, oracle_and_pyref, compare_load

SEC = cdefs.FLAG_SECONDARY
REV = cdefs.FLAG_REVERSE
M2 = cdefs.FLAG_MATE2


# ---------------------------------------------------------------- k-mers
def test_kmers_known_answers():
    # A=0 C=1 G=2 T=3; fw("ACG") = 0b000110 = 6, rc = "CGT" = 0b011011 = 27 -> canonical 6 (kmers.rs:192-196)
    assert O.kmers(b"ACG", 3) == [6]
    assert O.kmers(b"ACG", 3, canonical=False) == [6]
    # "CGT" is the reverse complement of "ACG"
    assert O.kmers(b"CGT", 3) == [6] and O.kmers(b"CGT", 3, canonical=False) == [27]
    # palindrome ACGT: both strands equal
    assert O.kmers(b"ACGT", 4) == [0b00011011]
    undef = (1 << 128) - 1
    # an N poisons every window that contains it (kmers.rs:184-190, 198-199)
    got = O.kmers(b"ACNGTAC", 3)
    assert got[:3] == [undef, undef, undef] and got[3] == O.kmers(b"GTA", 3)[0] and got[4] == O.kmers(b"TAC", 3)[0]
    assert O.kmers(b"acgt", 2) == [undef] * 3       # lower case is not ACGT
    assert O.kmers(b"AC", 3) == []


def test_kmers_match_python_transliteration():
    rng = np.random.default_rng(3)
    for k in (2, 5, 25, 31, 33, 63):
        s = bytes(rng.choice(list(b"ACGTN"), size=300, p=[0.24, 0.24, 0.24, 0.24, 0.04]).astype(np.uint8))
        assert O.kmers(s, k) == pyref.kmers(s, k)
        assert O.kmers(s, k, canonical=False) == pyref.kmers(s, k, canonical=False)


def test_linguistic_complexity():
    rng = np.random.default_rng(4)
    s = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 700)])
    s = s[:200] + b"A" * 120 + b"ACAC" * 30 + s[440:] + b"N" + s[:50]
    got = O.complexity_counts(s, 5, 300)
    assert list(got) == pyref.complexity_counts(s, 5, 300)
    # brute force: distinct 5-mer codes per window (UNDEF counts as one value)
    km = pyref.kmers(s, 5, canonical=False, bits=32)
    brute = [len(set(km[i:i + 296])) for i in range(len(s) - 300 + 1)]
    assert list(got) == brute
    assert got[200 - 0] < 296 and min(got) >= 1


# ---------------------------------------------------------------- genotypes / prefilter
def test_genotype_order_and_count():
    g = O.generate_genotypes(4, 2)
    assert g.tolist() == [[0, 0], [0, 1], [0, 2], [0, 3], [1, 1], [1, 2], [1, 3], [2, 2], [2, 3], [3, 3]]
    assert O.generate_genotypes(3, 3).tolist() == [list(t) for t in pyref.gen_combinations_with_repl(3, 3)]
    assert O.lib().orc_count_genotypes(256, 2) == 32896 and O.lib().orc_count_genotypes(4096, 2) == 8390656
    assert O.lib().orc_count_genotypes(5, 1) == 5 and O.lib().orc_count_genotypes(8, 3) == 120


def test_run_filter_known_answer():
    # 3 alleles x 4 reads; score(g) = prior + sum_r max over alleles (solve.rs:105-119)
    M = np.array([[-1.0, -5.0, -2.0, -9.0],
                  [-3.0, -1.0, -2.5, -9.0],
                  [-7.0, -7.0, -0.5, -1.0]])
    g = O.generate_genotypes(3, 2)
    sc = O.run_filter(M, g)
    want = {(0, 0): -17.0, (0, 1): -13.0, (0, 2): -7.5, (1, 1): -15.5, (1, 2): -5.5, (2, 2): -15.5}
    for ids, s in zip(g.tolist(), sc):
        assert s == want[tuple(ids)]
    pri = np.array([0.0, -1.0, -2.0, -3.0, -4.0, -5.0])
    assert np.array_equal(O.run_filter(M, g, pri), sc + pri)
    assert O.run_filter(M, g).tolist() == pyref.run_filter(M.tolist(), g.tolist())


def test_truncate_ixs_cases():
    sc = np.array([-10.0, -500.0, -12.0, -300.0, -11.0, -250.0])
    ix = np.arange(6)
    # threshold keeps scores >= best - 100
    assert O.truncate(sc, ix, 100.0, 1, 1).tolist() == [0, 4, 2]
    # min_size raises the kept count to the score of the min_size-th genotype (ties included)
    assert O.truncate(sc, ix, 100.0, 4, 1).tolist() == [0, 4, 2, 5]
    # at least `threads`
    assert O.truncate(sc, ix, 0.5, 1, 5).tolist() == [0, 4, 2, 5, 3]
    # min_size >= n or worst within threshold: keep all, sorted
    assert O.truncate(sc, ix, 100.0, 6, 1).tolist() == [0, 4, 2, 5, 3, 1]
    assert O.truncate(sc, ix, 1000.0, 1, 1).tolist() == [0, 4, 2, 5, 3, 1]
    ties = np.array([-1.0, -1.0, -1.0, -50.0, -50.0, -900.0])
    assert O.truncate(ties, ix, 10.0, 4, 1).tolist() == [0, 1, 2, 3, 4]      # tie at the min_size boundary kept
    rng = np.random.default_rng(0)
    for _ in range(20):
        s = np.round(rng.normal(-1000, 200, 40), 0)
        fd, ms, th = float(rng.integers(1, 500)), int(rng.integers(1, 45)), int(rng.integers(1, 10))
        assert O.truncate(s, np.arange(40), fd, ms, th).tolist() == pyref.truncate_ixs(s.tolist(), list(range(40)), fd, ms, th)


# ---------------------------------------------------------------- AllAlignments::load
def _setup(n_alleles=3, length=1200, paired=True, tech=cdefs.TECH_ILLUMINA, **prm):
    alleles = random_alleles(n_alleles, length, seed=11)
    bg = make_bg(technology=tech, paired=paired)
    p = O.default_params()
    for k, v in prm.items():
        setattr(p, k, v)
    O.resolve_params(p, bg)
    return alleles, bg, p


def test_load_single_perfect_pair_known_answer():
    alleles, bg, p = _setup()
    ol, pl, _ = oracle_and_pyref(alleles, 25, bg, p)
    a0 = alleles[0]
    s1, s2 = a0[300:450], a0[600:750]
    ch = ReadsChunk.from_pairs([{"seq1": s1.decode(), "seq2": s2.decode(),
                                 "recs": [(0, 300, 0, "150="), (0, 600, M2 | REV, "150=")]}])
    oa = ol.load(ch)
    assert oa.status.tolist() == [cdefs.READ_GOOD]
    # every k-mer is locus unique (all counts 0): 126 k-mers, non-overlapping greedy -> ceil(126/25) = 6 per mate
    assert oa.uniq_kmers.tolist() == [6, 6]
    assert oa.weight[0] == 1.0
    # one pair alignment: normalised ln-probs are 0 + 0 + ln NB(insert = 750 - 300)
    ins = ol.insert_lnprob(450)
    pen = ol.insert_penalty()
    assert oa.pair_alns["ln_prob"].tolist() == [ins]
    assert (int(oa.pair_alns["mid1"][0]), int(oa.pair_alns["mid2"][0])) == (375, 675)
    assert oa.unmapped_prob[0] == 1.0 * (2.0 * p.unmapped_penalty + pen)
    M = oa.best_aln_matrix()
    assert M[:, 0].tolist() == [ins, oa.unmapped_prob[0], oa.unmapped_prob[0]]
    compare_load(oa, pyref.load(pl, ch))


def test_load_edge_cases_against_pyref():
    alleles, bg, p = _setup(length=2600)
    ol, pl, _ = oracle_and_pyref(alleles, 25, bg, p)
    a0, a1 = alleles[0], alleles[1]
    s1, s2 = a0[300:450].decode(), a0[620:770].decode()
    pairs = [
        # unmapped first mate -> poorly mapped
        {"seq1": s1, "seq2": s2, "recs": [(0, 0, cdefs.FLAG_UNMAPPED, ""), (0, 620, M2 | REV, "150=")]},
        # unmapped second mate
        {"seq1": s1, "seq2": s2, "recs": [(0, 300, 0, "150="), (0, 0, M2 | cdefs.FLAG_UNMAPPED, "")]},
        # primary with too many mismatches (edit 12 > passable 9) -> rejected before the secondaries are read
        {"seq1": s1, "seq2": s2, "recs": [(0, 300, 0, "100=12X38="), (1, 300, SEC, "150="), (0, 620, M2 | REV, "150=")]},
        # best edit between good (4) and passable (9): rejected (threshold = good), weight untouched
        {"seq1": s1, "seq2": s2, "recs": [(0, 300, 0, "100=6X44="), (0, 620, M2 | REV, "150=")]},
        # secondary better than primary, hard clip on a secondary becomes soft, soft clip counted in edit
        {"seq1": s1, "seq2": s2, "recs": [(0, 300, 0, "100=3X47="), (1, 300, SEC, "150="), (2, 302, SEC, "2H148="),
                                          (0, 620, M2 | REV, "147=3S"), (1, 620, M2 | REV | SEC, "150=")]},
        # duplicates in one 128-bp bin: the better one wins, equal ones keep the first; second bin separate
        {"seq1": s1, "seq2": s2, "recs": [(0, 300, 0, "149=1X"), (0, 301, SEC, "150="), (0, 303, SEC, "150="),
                                          (0, 390, SEC, "148=2X"),
                                          (0, 620, M2 | REV, "150="), (0, 621, M2 | REV | SEC, "150=")]},
        # same-strand mates: no pair, two "alone" alignments
        {"seq1": s1, "seq2": s2, "recs": [(0, 300, 0, "150="), (0, 620, M2, "150=")]},
        # insertion / deletion / clipping limited by the contig start (limited_clipping)
        {"seq1": s1, "seq2": s2, "recs": [(0, 1, 0, "2S73=1I30=1D44="), (0, 620, M2 | REV, "150=")]},
        # out of bounds: both mates in the boundary region
        {"seq1": s1, "seq2": s2, "recs": [(0, 0, 0, "150="), (0, 10, M2 | REV, "150=")]},
        # more than 10 alignments of one end on one contig (distinct bins) -> top-10 kept
        {"seq1": s1, "seq2": s2, "recs": [(0, 300, 0, "150=")] +
            [(1, 128 * i + 3, SEC | (REV if i % 3 == 0 else 0), "149=1X" if i % 2 else "150=") for i in range(1, 15)] +
            [(0, 620, M2 | REV, "150=")]},
        # empty CIGAR on a secondary is skipped
        {"seq1": s1, "seq2": s2, "recs": [(0, 300, 0, "150="), (1, 300, SEC, ""), (0, 620, M2 | REV, "150=")]},
        # far apart mates: insert size beyond the cached range (direct NBinom evaluation)
        {"seq1": s1, "seq2": s2, "recs": [(0, 210, 0, "150="), (0, 1040, M2 | REV, "150=")]},
        # N bases in the read
        {"seq1": s1[:40] + "N" + s1[41:], "seq2": s2, "recs": [(0, 300, 0, "150="), (0, 620, M2 | REV, "150=")]},
    ]
    ch = ReadsChunk.from_pairs(pairs)
    oa = ol.load(ch)
    py = pyref.load(pl, ch)
    compare_load(oa, py)
    st = oa.status.tolist()
    assert st[0] == st[1] == st[2] == st[3] == cdefs.READ_POORLY_MAPPED
    assert st[8] == cdefs.READ_OUT_OF_BOUNDS
    assert st[4] == st[5] == st[6] == st[7] == st[9] == st[10] == st[11] == st[12] == cdefs.READ_GOOD
    # pair 5: dedupe keeps rec 1 (first of the two perfect ones) in bin 2 and rec 3 in bin 3
    lo, hi = int(oa.pa_off[5]), int(oa.pa_off[6])
    assert sorted(set(oa.pair_alns["ix1"][lo:hi].tolist())) == [1, 3]
    assert set(oa.pair_alns["ix2"][lo:hi].tolist()) == {4}
    # pair 6: two alone entries, no joint one
    lo, hi = int(oa.pa_off[6]), int(oa.pa_off[7])
    pa = oa.pair_alns[lo:hi]
    assert len(pa) == 2 and {(int(x["ix1"]), int(x["ix2"])) for x in pa} == {(0, cdefs.NONE_U32), (cdefs.NONE_U32, 1)}
    # pair 9: at most 10 entries on contig 1
    lo, hi = int(oa.pa_off[9]), int(oa.pa_off[10])
    assert (oa.pair_alns["contig"][lo:hi] == 1).sum() == 10
    # N reduces the unique k-mer count of mate 1 (windows with N are UNDEF)
    assert oa.uniq_kmers[2 * 12] < 6 and oa.uniq_kmers[2 * 12 + 1] == 6


def test_load_invalid_inputs_raise():
    alleles, bg, p = _setup()
    ol, pl, _ = oracle_and_pyref(alleles, 25, bg, p)
    s = alleles[0][300:450].decode()
    bad = [
        [(0, 300, 0, "150M"), (0, 620, M2 | REV, "150=")],            # M needs --eqx (aln.rs:311)
        [(0, 300, 0, "3H147="), (0, 620, M2 | REV, "150=")],          # hard-clipped primary (locs.rs:526)
        [(0, 300, 0, "150=")],                                        # paired data without a second mate
        [(7, 300, 0, "150="), (0, 620, M2 | REV, "150=")],            # unknown contig
        [],                                                           # no records at all
    ]
    for recs in bad:
        ch = ReadsChunk.from_pairs([{"seq1": s, "seq2": s, "recs": recs}])
        with pytest.raises(ValueError):
            ol.load(ch)
        with pytest.raises(pyref.InvalidData):
            pyref.load(pl, ch)
    # the same defects in records that the reference never examines are harmless
    ch = ReadsChunk.from_pairs([{"seq1": s, "seq2": s, "recs": [(0, 0, cdefs.FLAG_UNMAPPED, ""), (0, 620, M2 | REV, "150M")]}])
    assert ol.load(ch).status.tolist() == [cdefs.READ_POORLY_MAPPED]


def test_load_empty_batch():
    alleles, bg, p = _setup()
    ol, pl, _ = oracle_and_pyref(alleles, 25, bg, p)
    oa = ol.load(ReadsChunk.from_pairs([]))
    assert oa.n_pairs == 0 and oa.n_good == 0 and oa.best_aln_matrix().shape == (3, 0)


@pytest.mark.parametrize("tech,paired,rl", [(cdefs.TECH_ILLUMINA, True, 150), (cdefs.TECH_NANOPORE, False, 1500)])
def test_load_synthetic_against_pyref(tech, paired, rl):
    L = synth.SynthLocus(6, 400, seed=77, base_len=6000, technology=tech, read_len=rl)
    p = O.resolve_params(O.default_params(), L.bg)
    alleles = [L.allele(a) for a in range(6)]
    counts = [L.counts[int(L.cnt_off[a]):int(L.cnt_off[a + 1])] for a in range(6)]
    ol, pl, _ = oracle_and_pyref(alleles, L.k, L.bg, p, counts)
    ch = L.reads(0, 120)
    oa = ol.load(ch)
    compare_load(oa, pyref.load(pl, ch))
    assert oa.n_good > 60
    M = oa.best_aln_matrix()
    gts = O.generate_genotypes(6, 2)
    sc = O.run_filter(M, gts)
    assert tuple(gts[int(np.argmax(sc))]) == L.true_genotype


def test_strict_subset_and_poor_complexity():
    # low-complexity allele region relaxes the thresholds (locs.rs:533-536)
    alleles = random_alleles(2, 1500, seed=5)
    alleles[0] = alleles[0][:400] + b"AC" * 250 + alleles[0][900:]
    bg = make_bg()
    p = O.resolve_params(O.default_params(), bg)
    ol, pl, _ = oracle_and_pyref(alleles, 25, bg, p)
    s = alleles[0][500:650].decode()
    pairs = [{"seq1": s, "seq2": s, "recs": [(0, 500, 0, "100=30X20="), (0, 700, M2 | REV, "150=")]}]
    ch = ReadsChunk.from_pairs(pairs)
    oa = ol.load(ch)
    compare_load(oa, pyref.load(pl, ch))
    assert oa.status[0] != cdefs.READ_POORLY_MAPPED          # threshold = floor(0.7 * 150) = 105 >= 30
    assert abs(oa.weight[0] - math.sqrt(4.0 / 30.0) * 1.0) < 1e-15 or oa.weight[0] < 1.0
    p2 = O.resolve_params(O.default_params(), bg)
    p2.strict_subset = 1
    ol2, pl2, _ = oracle_and_pyref(alleles, 25, bg, p2)
    s0 = alleles[1][200:350].decode()
    ch2 = ReadsChunk.from_pairs([{"seq1": s0, "seq2": s0, "recs": [(1, 200, 0, "140=6X4="), (1, 500, M2 | REV, "150=")]}])
    oa2 = ol2.load(ch2)
    compare_load(oa2, pyref.load(pl2, ch2))
    # passes read_next_alns (passable) but fails best_edit_is_good (locs.rs:1262-1265)
    assert oa2.status[0] == cdefs.READ_POORLY_MAPPED
