"""Explicit region weights (`--reg-weights`; model/windows.rs:196-317, 409-413, 443, 493-503, 683-693; locs.rs:860, 903):
known answers of the fixed-point averages, the loader's errors, and oracle (C) == independent Python transliteration
bit for bit. CPU only."""
import ctypes as C

import numpy as np
import pytest

from locityper_amd import cdefs, synth
from tests import oracle_ffi as O
from tests import pyref
from tests.helpers import make_bg, random_alleles, oracle_and_pyref, compare_load


def bed_lines(lengths, rng, piece=(40, 400), values=(0.0, 0.1, 0.25, 1.0 / 3.0, 0.5, 0.9, 1.0)):
    """Random full coverage of every allele, as parsed lines (allele, start, end, value) in file order."""
    lines = []
    for a, n in enumerate(lengths):
        pos = 0
        while pos < n:
            e = min(n, pos + int(rng.integers(piece[0], piece[1])))
            lines.append((a, pos, e, float(values[int(rng.integers(0, len(values)))])))
            pos = e
    return lines


def columns(lines):
    return [np.array([x[i] for x in lines], dtype=t) for i, t in enumerate((np.uint32, np.uint32, np.uint32, np.float64))]


def test_fixed_point_average_known_answers():
    # sums are kept in units of 2^-32 (windows.rs:204, 212-218) and the average divides the INTEGER sum (236-238)
    w = pyref.PyExplicitWeights()
    w.extend_by(2, 0.5); w.extend_by(1, 1.0); w.finish()
    assert [x[1] for x in w.weights] == [0, 1 << 31, 1 << 32, 1 << 33]
    assert w.average(0, 2) == 0.5 and w.average(0, 3) == float(((1 << 33) // 3)) / 2.0 ** 32      # 2/3 rounded DOWN to 2^-32
    assert w.average(0, 3) < 2.0 / 3.0 and w.at(3) == 1.0                                           # finish() repeats the last value
    v = pyref.PyExplicitWeights()
    v.extend_by(3, 1.0 / 3.0); v.finish()
    third = int((1.0 / 3.0) * 2.0 ** 32)
    assert v.weights[3][1] == 3 * third and v.average(0, 3) == third / 2.0 ** 32 != 1.0 / 3.0


def _locus(n_alleles=3, length=1500, paired=True):
    alleles = random_alleles(n_alleles, length, seed=21)
    alleles[1] = alleles[1][:-37]                       # different lengths
    bg = make_bg(paired=paired)
    p = O.resolve_params(O.default_params(), bg)
    ol, pl, _ = oracle_and_pyref(alleles, 25, bg, p)
    return alleles, bg, p, ol, pl


def test_loader_errors_as_upstream():
    alleles, bg, p, ol, pl = _locus()
    n = [len(a) for a in alleles]
    ok = [(0, 0, n[0], 1.0), (1, 0, 100, 0.5), (1, 100, n[1], 0.25), (2, 0, n[2], 0.0)]

    def both(lines):
        rc = ol.set_explicit_weights(*columns(lines))
        try:
            pl.set_explicit_weights(lines)
            py = 0
        except pyref.ParsingError:
            py = cdefs.ERR_INVALID_DATA
        except ValueError:
            py = cdefs.ERR_INVALID_INPUT
        assert rc == py, (rc, py)
        return rc
    assert both(ok) == 0
    assert both(ok + [(7, 0, 10, 0.5)]) == 0                                   # unknown contig: line ignored (269-272)
    assert both(ok[:3]) == cdefs.ERR_INVALID_DATA                              # haplotype missing (305-308)
    assert both([(0, 0, n[0] - 1, 1.0)] + ok[1:]) == cdefs.ERR_INVALID_DATA    # not fully covered (309-313)
    assert both([(0, 5, n[0], 1.0)] + ok[1:]) == cdefs.ERR_INVALID_DATA        # does not start where the last one ended (291-295)
    assert both([ok[0], (1, 0, 100, 0.5), (1, 90, n[1], 0.25), ok[3]]) == cdefs.ERR_INVALID_DATA     # overlap
    assert both([(0, 0, n[0], 1.5)] + ok[1:]) == cdefs.ERR_INVALID_DATA        # value outside [0, 1] (285-288)
    assert both([(0, 0, n[0] + 1, 1.0)] + ok[1:]) == cdefs.ERR_INVALID_INPUT   # beyond the allele (interv.rs:112-116)
    # interleaved alleles are fine as long as each one is covered in order
    assert both([ok[1], ok[0], ok[3], ok[2]]) == 0


def test_window_weight_carries_the_window_average():
    alleles, bg, p, ol, pl = _locus()
    rng = np.random.default_rng(3)
    lines = bed_lines([len(a) for a in alleles], rng)
    info = [ol.contig_info(a) for a in range(3)]
    plain = {}
    for a in range(3):
        for j in range(info[a][3]):
            ws = info[a][4] + j * bg.window
            plain[a, j] = O.lib().orc_window_weight(ol._h, a, ws, None)
    assert ol.set_explicit_weights(*columns(lines)) == 0
    pl.set_explicit_weights(lines)
    per_base = [np.concatenate([np.full(e - s, v) for (aa, s, e, v) in lines if aa == a]) for a in range(3)]
    left_padding = (bg.neighb - bg.window) // 2
    for (a, j), w0 in plain.items():
        ws = info[a][4] + j * bg.window
        i = ws - left_padding
        ew = pl.window_explicit_weight(a, i)
        assert O.lib().orc_window_weight(ol._h, a, ws, None) == w0 * ew          # windows.rs:441-443: the last factor
        # the window average covers exactly the window's own bases (409-413), up to the 2^-32 truncation
        assert 0.0 <= per_base[a][ws:ws + bg.window].mean() - ew < bg.window * 2.0 ** -32 + 1e-12


@pytest.mark.parametrize("tech,paired,rl", [(cdefs.TECH_ILLUMINA, True, 150), (cdefs.TECH_NANOPORE, False, 1500)])
def test_load_with_explicit_weights_against_pyref(tech, paired, rl):
    L = synth.SynthLocus(6, 400, seed=78, base_len=6000, technology=tech, read_len=rl)
    p = O.resolve_params(O.default_params(), L.bg)
    alleles = [L.allele(a) for a in range(6)]
    counts = [L.counts[int(L.cnt_off[a]):int(L.cnt_off[a + 1])] for a in range(6)]
    ol, pl, _ = oracle_and_pyref(alleles, L.k, L.bg, p, counts)
    ch = L.reads(0, 120)
    plain = ol.load(ch)
    lines = bed_lines([len(a) for a in alleles], np.random.default_rng(5), piece=(30, 900))
    assert ol.set_explicit_weights(*columns(lines)) == 0
    pl.set_explicit_weights(lines)
    oa = ol.load(ch)
    compare_load(oa, pyref.load(pl, ch))
    # the explicit weight only scales: same pair alignments, weight = plain weight x mean of the larger read-end weights
    both = np.nonzero((plain.status == cdefs.READ_GOOD) | (plain.status == cdefs.READ_FEW_KMERS))[0]
    assert len(both) > 60
    scaled = 0
    for r in both:
        lo, hi = int(plain.pa_off[r]), int(plain.pa_off[r + 1])
        assert hi - lo == int(oa.pa_off[r + 1] - oa.pa_off[r])
        ew = pl.explicit_read_weight([(0.0, int(x["contig"]), 0, int(x["mid1"]), 0, int(x["mid2"])) for x in plain.pair_alns[lo:hi]])
        assert 0.0 <= ew <= 1.0 and oa.weight[r] == plain.weight[r] * ew
        assert oa.status[r] == (cdefs.READ_GOOD if oa.weight[r] >= p.min_weight else cdefs.READ_FEW_KMERS)
        scaled += ew < 1.0
    assert scaled > 30
    # a file of ones changes nothing at all
    ones = [(a, 0, len(alleles[a]), 1.0) for a in range(6)]
    assert ol.set_explicit_weights(*columns(ones)) == 0
    o1 = ol.load(ch)
    assert np.array_equal(o1.weight, plain.weight) and np.array_equal(o1.pair_alns["ln_prob"], plain.pair_alns["ln_prob"])
    # a region of zeros takes the reads whose every location lies inside it out of the analysis (weight 0 < min_weight)
    zeros = [(a, 0, len(alleles[a]), 0.0) for a in range(6)]
    assert ol.set_explicit_weights(*columns(zeros)) == 0
    oz = ol.load(ch)
    assert oz.n_good == 0 and not np.any(oz.status == cdefs.READ_GOOD)


def test_read_end_weight_looks_half_a_window_to_either_side():
    # windows.rs:493-503: max over the middle and middle -+ window/2, clamped to [0, len] (len: the entry finish() appends)
    alleles, bg, p, ol, pl = _locus()
    n = [len(a) for a in alleles]
    lines = [(0, 0, 500, 0.1), (0, 500, 520, 0.9), (0, 520, n[0], 0.2), (1, 0, n[1], 1.0), (2, 0, n[2] - 1, 0.3), (2, n[2] - 1, n[2], 0.8)]
    pl.set_explicit_weights(lines)
    u = bg.window // 2
    assert pl.read_end_weight(0, 510) == 0.9 and pl.read_end_weight(0, 500 - u) == 0.9 and pl.read_end_weight(0, 519 + u) == 0.9
    assert pl.read_end_weight(0, 499 - u) == 0.1 and pl.read_end_weight(0, 520 + u) == 0.2
    assert pl.read_end_weight(0, 10) == 0.1                                   # saturating_sub
    assert pl.read_end_weight(2, n[2] - 10) == 0.8                            # min(i + u, n - 1) reaches the appended entry
    assert pl.read_end_weight(0, None) == 0.0 and pl.read_end_weight(0, cdefs.NONE_U32) == 0.0
    assert pl.explicit_read_weight([(0.0, 0, 0, 510, 0, cdefs.NONE_U32), (0.0, 0, 0, 10, 0, 700)]) == (0.9 + 0.2) / 2
