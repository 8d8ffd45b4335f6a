"""Oracle alignment recovery: the aligner against brute force, CigarIndex / transfer properties
(the transferred CIGAR consumes the read, its = / X columns are true, it is as good as a direct alignment wherever the
haplotypes differ by simple events), and AllAlignments::load with recovery against load with every alignment given."""
import itertools

import numpy as np
import pytest

from locityper_amd import api, cdefs
from locityper_amd.cdefs import ReadsChunk
from tests import oracle_ffi as O
from tests.helpers import make_bg, locus_arrays

X, GO, GE = 4, 6, 1


def score_of(cigar, match_bonus=0):
    pen, num = 0, ""
    for ch in cigar:
        if ch.isdigit(): num += ch; continue
        n = int(num); num = ""
        pen += {"=": -match_bonus * n, "X": X * n, "I": GO + GE * n, "D": GO + GE * n}[ch]
    return pen


def columns_true(cigar, ref, qry):
    i = j = 0; num = ""
    for ch in cigar:
        if ch.isdigit(): num += ch; continue
        n = int(num); num = ""
        if ch in "=X":
            for t in range(n):
                if (ref[i + t] == qry[j + t]) != (ch == "="): return False
            i += n; j += n
        elif ch in "IS": j += n
        elif ch == "D": i += n
    return (i, j)


def brute_global(a, b):
    """Optimal gap-affine penalty by plain dynamic programming in Python (three matrices)."""
    INF = 10 ** 9
    n, m = len(a), len(b)
    M = [[INF] * (m + 1) for _ in range(n + 1)]; D = [[INF] * (m + 1) for _ in range(n + 1)]; I = [[INF] * (m + 1) for _ in range(n + 1)]
    M[0][0] = 0
    for i in range(n + 1):
        for j in range(m + 1):
            if i and j: M[i][j] = min(M[i - 1][j - 1], D[i - 1][j - 1], I[i - 1][j - 1]) + (0 if a[i - 1] == b[j - 1] else X)
            if i: D[i][j] = min(min(M[i - 1][j], I[i - 1][j]) + GO + GE, D[i - 1][j] + GE)
            if j: I[i][j] = min(min(M[i][j - 1], D[i][j - 1]) + GO + GE, I[i][j - 1] + GE)
    return min(M[n][m], D[n][m], I[n][m])


def test_aligner_is_optimal_and_consistent():
    rng = np.random.default_rng(3)
    for _ in range(60):
        n = int(rng.integers(1, 40)); a = bytes(rng.choice(list(b"ACGT"), n).tolist())
        b = bytearray(a)
        for _ in range(int(rng.integers(0, 5))):                      # a few random edits
            p = int(rng.integers(0, max(1, len(b))))
            r = rng.random()
            if r < 0.4 and b: b[p % len(b)] = int(rng.choice(list(b"ACGT")))
            elif r < 0.7: b[p:p] = bytes(rng.choice(list(b"ACGT"), int(rng.integers(1, 4))).tolist())
            elif len(b) > 3: del b[p % len(b):(p % len(b)) + int(rng.integers(1, 3))]
        b = bytes(b) or b"A"
        pen, cg = O.dp_align(a, b)
        assert pen == brute_global(a, b) == score_of(cg)
        assert columns_true(cg, a, b) == (len(a), len(b))
    # known answers: a mismatch is cheaper than two gaps; a long gap is one event
    assert O.dp_align(b"ACGTACGT", b"ACGAACGT") == (4, "3=1X4=")
    assert O.dp_align(b"ACGTTTTTACGT", b"ACGTACGT")[0] == GO + 4 * GE
    # free begin / free end with the match bonus of the semi-global aligner (wfa.rs:194-197)
    pen, cg = O.dp_align(b"TTTTTACGTACGT", b"GGACGTACGT", 2, 1)            # the query's GG and the reference's TTTTT are skipped
    assert cg.endswith("8=") and columns_true(cg, b"TTTTTACGTACGT", b"GGACGTACGT") == (13, 10)
    pen, cg = O.dp_align(b"ACGTACGTCC", b"ACGTACGTAAAA", 2, 2)
    assert cg.startswith("8=") and columns_true(cg, b"ACGTACGTCC", b"ACGTACGTAAAA") == (10, 12)


def make_haps(rng, n, length):
    base = rng.choice(list(b"ACGT"), length).astype(np.uint8)
    haps = []
    for _ in range(n):
        s = bytearray(base.tobytes())
        pos = sorted(rng.choice(np.arange(60, length - 60), int(length * 0.012), replace=False).tolist(), reverse=True)
        for p in pos:
            r = rng.random()
            if r < 0.7: s[p] = int(rng.choice([c for c in b"ACGT" if c != s[p]]))
            elif r < 0.85: s[p:p] = bytes(rng.choice(list(b"ACGT"), int(rng.integers(1, 9))).tolist())
            else: del s[p:p + int(rng.integers(1, 9))]
        haps.append(bytes(s))
    return haps


def hap_alns_for(haps, transfer_fails=3):
    H = O.HapAlns(len(haps), transfer_fails, 0.2)
    for i, j in itertools.combinations(range(len(haps)), 2):
        pen, cg = O.dp_align(haps[j], haps[i])                           # query i on reference j
        H.add(i, j, cg)
    H.sort()
    return H


def test_transfer_properties():
    # upstream's optimize() step realigns against the wrong stretch of the target for reads (see lcty_oracle_transfer.c);
    # the properties below are those of everything before it
    O.lib().orc_transfer_set_optimize(0)
    try:
        _transfer_properties()
    finally:
        O.lib().orc_transfer_set_optimize(1)


def test_optimize_step_is_kept_as_written_upstream():
    rng = np.random.default_rng(11)
    haps = make_haps(rng, 4, 1800)
    H = hap_alns_for(haps)
    # the read ends inside a 3-base insertion of haplotype 1: tail "3D3=1I" after the last anchor = insertion + deletion
    O.lib().orc_transfer_set_optimize(0)
    st0, cg0 = H.transfer_one(1, 2, 399, "150=", haps[1][399:549], haps[2])
    O.lib().orc_transfer_set_optimize(1)
    st1, cg1 = H.transfer_one(1, 2, 399, "150=", haps[1][399:549], haps[2])
    assert (st0, cg0) == (391, "49=1X54=1X33=1X7=3D3=1S") and st1 == 391 and cg1 != cg0
    # ... and the realigned tail is the global alignment of the read's tail with target[146..152), not target[537..543)
    pen, tail = O.dp_align(haps[2][146:152], haps[1][399:549][146:150])
    assert cg1 == "49=1X54=1X33=1X" + O.cigar_str(O.cigar_words("7=" + tail)).replace("7=4=", "11=")


def _transfer_properties():
    rng = np.random.default_rng(11)
    haps = make_haps(rng, 4, 1800)
    H = hap_alns_for(haps)
    n_checked = n_equal_best = 0
    for trial in range(150):
        src, dst = rng.choice(4, 2, replace=False).tolist()
        start = int(rng.integers(40, len(haps[src]) - 200))
        read = bytearray(haps[src][start:start + 150])
        if rng.random() < 0.5:                                           # a sequencing error or two
            for _ in range(int(rng.integers(1, 3))):
                p = int(rng.integers(0, 150)); read[p] = int(rng.choice([c for c in b"ACGT" if c != read[p]]))
        read = bytes(read)
        pen, read_cigar = O.dp_align(haps[src][start:start + 150], read)
        new_start, cg = H.transfer_one(src, dst, start, read_cigar, read, haps[dst])
        # consumes the whole read; every = / X column is what it says
        end = columns_true(cg.replace("S", "I"), haps[dst][new_start:], read)
        assert end and end[1] == 150, (trial, cg)
        # soft clips only at the ends, no insertion left at the ends
        assert "S" not in cg[1:-1].strip("0123456789")[1:-1] or cg.count("S") <= 2
        assert not cg.lstrip("0123456789").startswith("I") and not cg.endswith("I")
        # as good as aligning the read directly to the stretch it lands on (when nothing was clipped)
        if "S" not in cg:
            direct, _ = O.dp_align(haps[dst][new_start:new_start + end[0]], read)
            n_checked += 1
            n_equal_best += score_of(cg) == direct
            assert score_of(cg) <= direct + 2 * (GO + GE), (trial, cg, direct)
    assert n_checked > 80 and n_equal_best >= 0.9 * n_checked
    # a read inside a long identical stretch is copied with its CIGAR (FULL_MATCH_PADDING, cigar.rs:1281-1286)
    same = O.HapAlns(2, 3, 0.2)
    a = haps[0]
    same.add(0, 1, f"{len(a)}=")
    same.sort()
    assert same.transfer_one(0, 1, 500, "100=1X49=", a[500:650], a) == (500, "100=1X49=")


def test_load_with_recovery_against_full_alignment_lists():
    """Give the mapper's full answer (an alignment on every allele) to load(), and only the primary alignments to
    load_recover(): the recovered per-allele likelihood rows must match wherever a transferred alignment exists."""
    rng = np.random.default_rng(5)
    haps = make_haps(rng, 5, 2600)
    bg = make_bg()
    p = O.resolve_params(O.default_params(), bg)
    seqs, seq_off, cflat, cnt_off, _ = locus_arrays([bytearray(h) for h in haps], 25)
    ol = O.OracleLocus(seqs, seq_off, cflat, cnt_off, 25, bg, p)
    H = hap_alns_for(haps, transfer_fails=3)
    M2, REV = cdefs.FLAG_MATE2, cdefs.FLAG_REVERSE
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    pairs = []
    for _ in range(120):
        src = int(rng.integers(0, 5))
        p1 = int(rng.integers(320, len(haps[src]) - 800)); p2 = p1 + int(rng.integers(200, 420))
        r1, r2 = haps[src][p1:p1 + 150], haps[src][p2:p2 + 150]
        # BAM stores the reverse-strand mate as the reference strand; the flat table keeps the primary record's SEQ
        pairs.append({"seq1": r1.decode(), "seq2": r2.decode(), "recs": [(src, p1, 0, "150="), (src, p2, M2 | REV, "150=")]})
    ch = ReadsChunk.from_pairs(pairs)
    plain = ol.load(ch)
    rec = ol.load_recover(ch, H)
    assert np.array_equal(plain.status, rec.status)
    Mp, Mr = plain.best_aln_matrix(), rec.best_aln_matrix()
    good = np.flatnonzero(plain.status == cdefs.READ_GOOD)
    assert len(good) > 80
    off_p, off_r = plain.pa_off.astype(np.int64), rec.pa_off.astype(np.int64)
    assert np.all(np.diff(off_r) >= np.diff(off_p)) and np.diff(off_r).sum() > 3 * np.diff(off_p).sum()
    # recovered rows: the source allele keeps its value; on the other alleles "no alignment = unmapped probability" is
    # replaced by the probability of the transferred pair (which may be higher or lower)
    src_col = np.array([p["recs"][0][0] for p in pairs])[good]
    assert np.allclose(Mr[src_col, np.arange(len(good))], Mp[src_col, np.arange(len(good))], rtol=0, atol=1e-9)
    changed = np.abs(Mr - Mp) > 1e-9
    assert changed.sum() > 3 * len(good) and (Mr > Mp + 1e-9).sum() > len(good)
    # deterministic
    rec2 = ol.load_recover(ch, H)
    assert np.array_equal(rec2.best_aln_matrix(), Mr)


def test_transfer_matches_the_independent_transliteration():
    """oracle/lcty_oracle_transfer.c against tests/pyref_transfer.py (written separately from the Rust): CigarIndex, the two-CIGAR
    walk, anchors, clipped ends, optimize — new start and CIGAR must be identical, for reads with substitutions and indels, soft
    clips, both directions of the haplotype alignment, reads at the ends of the alleles."""
    from tests import pyref_transfer as PT
    from tests.helpers import noisy_read
    rng = np.random.default_rng(17)
    haps = make_haps(rng, 4, 2000)
    H = hap_alns_for(haps)
    cigs, idxs = {}, {}
    for q, r, w, nm, ln in H.entries:                                  # query = lower id
        cigs[(q, r)] = PT.Cig.parse(O.cigar_str(w)); idxs[(q, r)] = PT.CigarIndex(cigs[(q, r)])
    n_copy = n_walk = n_clipped = 0
    for trial in range(400):
        src, dst = rng.choice(4, 2, replace=False).tolist()
        ln = int(rng.integers(60, 260))
        kind = trial % 4
        if kind == 3: start = int(rng.integers(0, 30)) if trial % 8 == 3 else len(haps[src]) - ln - int(rng.integers(0, 30))
        else: start = int(rng.integers(30, len(haps[src]) - ln - 40))
        read, cg = noisy_read(rng, haps[src], start, ln, err=0.0 if kind == 0 else 0.04)
        if kind == 2:                                                      # adapter at one end: soft clip
            k = int(rng.integers(3, 25))
            first = PT.Cig.parse(cg).t[0]
            if first[0] == "=" and first[1] > k + 5:
                read = bytes(rng.choice(list(b"ACGT"), k).tolist()) + read[k:]
                cg = f"{k}S{first[1] - k}=" + cg[len(str(first[1])) + 1:]
                start += k
        lo, hi = min(src, dst), max(src, dst)
        got_start, got = H.transfer_one(src, dst, start, cg, read, haps[dst])
        exp_start, exp = PT.transfer_read_alignment(cigs[(lo, hi)], idxs[(lo, hi)], src > dst, start, PT.Cig.parse(cg), read, haps[dst])
        assert (got_start, got) == (exp_start, str(exp)), (trial, src, dst, start, cg)
        n_copy += got == cg; n_walk += got != cg; n_clipped += "S" in got
    assert n_copy > 20 and n_walk > 150 and n_clipped > 20, (n_copy, n_walk, n_clipped)
