"""Oracle of minimizer read recruitment against known answers and the independent Python transliteration."""
import numpy as np

from tests import oracle_ffi as O
from tests import pyref_recruit as PR
from tests.helpers import locus_arrays, noisy_read

COMP = bytes.maketrans(b"ACGT", b"TGCA")


def revcomp(s):
    return bytes(s).translate(COMP)[::-1]


def test_hash_fraction_and_parameters():
    # fasthash mix (kmers.rs:93-103): worked by hand for 0: !0 = 2^64-1; ^= >>23; * 0x2127599bf4325c37; ^= >>47
    x = (1 << 64) - 1; x ^= x >> 23; x = (x * 0x2127599bf4325c37) & ((1 << 64) - 1); x ^= x >> 47
    assert O.lib().orc_fast_hash64(0) == x == PR.fast_hash(0)
    for v in (1, 12345678901234567, (1 << 30) - 1): assert O.lib().orc_fast_hash64(v) == PR.fast_hash(v)
    # continued fractions (frac.rs:50-76)
    assert O.fraction_approximate_u16(0.5) == (1, 2) == PR.approximate_u16(0.5)
    assert O.fraction_approximate_u16(0.7) == (7, 10) == PR.approximate_u16(0.7)
    assert O.fraction_approximate_u16(0.25) == (1, 4)
    n, d = O.fraction_approximate_u16(0.3333); assert abs(n / d - 0.3333) < 1e-6 and (n, d) == PR.approximate_u16(0.3333)
    # Params::new (recruit.rs:65-105): 2L / (w + 1) minimizers per stretch, score of a stretch at match_frac
    t = O.OracleTargets(15, 10, 0.5, 2000, 50)
    assert t.params() == ((1, 2), 364, 364) and PR.Targets().stretch_minims == 364 and PR.Targets().stretch_score == 364
    assert O.OracleTargets(15, 10, 0.25, 2000, 50).params()[2] == 3            # never below SUBSUM_BONUS


def test_minimizers_against_brute_force_and_transliteration():
    rng = np.random.default_rng(3)
    for trial in range(60):
        n = int(rng.integers(1, 400)); k = int(rng.integers(3, 32)); w = int(rng.integers(2, 40))
        seq = bytearray(rng.choice(list(b"ACGT"), n).tolist())
        for _ in range(int(rng.integers(0, 4))): seq[int(rng.integers(0, n))] = ord("N")
        got = O.canon_minimizers(seq, k, w)
        assert got == PR.canon_minimizers(bytes(seq), k, w)
        # brute force: canonical k-mer hashes; every window of w consecutive k-mers whose minimum is defined contributes its leftmost minimum
        hs = []
        for i in range(n - k + 1):
            kmer = bytes(seq[i:i + k])
            if any(c not in b"ACGT" for c in kmer): hs.append(None); continue
            enc = lambda s: int("".join(str(b"ACGT".index(c)) for c in s), 4)
            f, r = enc(kmer), enc(revcomp(kmer))
            hs.append((PR.fast_hash(min(f, r)), not r < f))
        got_pos = {p for p, _, _ in got}
        for p, h, fw in got: assert hs[p] == (h, fw)
        if not any(h is None for h in hs):                                   # without N the rule is the plain one
            exp = set()
            for s in range(0, len(hs) - w + 1):
                win = hs[s:s + w]
                exp.add(s + min(range(w), key=lambda j: (win[j][0], j)))
            assert got_pos == exp, (trial, k, w)
    assert O.canon_minimizers(b"ACGT", 15, 10) == []                          # shorter than k + w - 1


def _loci(rng, n_loci=3, n_alleles=3, length=3000):
    loci = []
    for _ in range(n_loci):
        base = rng.choice(list(b"ACGT"), length).astype(np.uint8)
        alleles = []
        for _ in range(n_alleles):
            s = base.copy(); m = rng.random(length) < 0.01
            s[m] = rng.choice(list(b"ACGT"), int(m.sum()))
            alleles.append(bytes(s))
        loci.append(alleles)
    # locus 2 shares a stretch with locus 0 (a paralogous segment)
    loci[2] = [a[:1000] + loci[0][0][500:1500] + a[2000:] for a in loci[2]]
    return loci


def _targets(loci, base_k=25, **kw):
    ot, pt = O.OracleTargets(**kw), PR.Targets(**kw)
    rng = np.random.default_rng(9)
    for alleles in loci:
        counts = [np.where(rng.random(len(a) - base_k + 1) < 0.1, 80, 0).astype(np.uint16) for a in alleles]
        seqs = np.frombuffer(b"".join(alleles), dtype=np.uint8)
        seq_off = np.cumsum([0] + [len(a) for a in alleles]).astype(np.uint64)
        cnt_off = np.cumsum([0] + [len(c) for c in counts]).astype(np.uint64)
        ot.add_locus(seqs, seq_off, np.concatenate(counts), cnt_off, base_k)
        pt.add(alleles, counts, base_k)
    ot.finalize()
    return ot, pt


def test_targets_and_recruitment_match_the_transliteration():
    rng = np.random.default_rng(21)
    loci = _loci(rng)
    for kw in (dict(k=15, w=10, match_frac=0.5), dict(k=27, w=5, match_frac=0.7, thresh_kmer_count=50), dict(k=11, w=20, match_frac=0.3)):
        ot, pt = _targets(loci, **kw)
        exp = sorted((m, l, d, r) for m, v in pt.minim_to_loci.items() for l, d, r in v)
        assert ot.entries() == exp
        n_rec = n_multi = 0
        for trial in range(300):
            li = int(rng.integers(0, len(loci))); al = loci[li][int(rng.integers(0, 3))]
            kind = trial % 6
            if kind == 5: r1 = bytes(rng.choice(list(b"ACGT"), 150).tolist()); r2 = bytes(rng.choice(list(b"ACGT"), 150).tolist())
            else:
                p = int(rng.integers(0, len(al) - 700))
                r1, _ = noisy_read(rng, al, p, 150, err=0.02 * (kind % 3)); r2, _ = noisy_read(rng, al, p + 350, 150, err=0.02 * (kind % 3))
                r2 = revcomp(r2)
                if kind == 4: r1 = r1[:40] + b"N" + r1[41:]
                if trial % 2: r1, r2 = revcomp(r1), revcomp(r2)
            got = ot.recruit(r1, r2)
            assert got == pt.recruit(r1, r2), (kw, trial)
            assert ot.recruit(r1) == pt.recruit(r1)
            n_rec += bool(got); n_multi += len(got) > 1
            if kind == 5: assert got == []
            elif kind == 0 and kw["match_frac"] <= 0.5: assert li in got
        assert n_rec > 60 and n_multi > 3, (n_rec, n_multi)


def test_long_reads_match_the_transliteration():
    rng = np.random.default_rng(33)
    loci = _loci(rng, length=6000)
    ot, pt = _targets(loci, k=15, w=10, match_frac=0.5, match_length=2000)
    n_rec = 0
    for trial in range(60):
        li = int(rng.integers(0, 3)); al = loci[li][int(rng.integers(0, 3))]
        ln = int(rng.integers(600, 5000)); p = int(rng.integers(0, len(al) - ln))
        r, _ = noisy_read(rng, al, p, ln, err=0.03 * (trial % 4))
        if trial % 5 == 4: r = bytes(rng.choice(list(b"ACGT"), 3000).tolist()) + r[:700]      # mostly foreign sequence
        if trial % 2: r = revcomp(r)
        got = ot.recruit(r)
        assert got == pt.recruit(r), trial
        n_rec += bool(got)
    assert 20 < n_rec < 60
