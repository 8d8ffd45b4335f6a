"""Independent pure-Python transliteration of the reference scoring path, for SMALL inputs only.

A second restatement (written separately from oracle/lcty_oracle.c, directly from the cited
reference lines) used to cross-check the C oracle: two independent transcriptions agreeing on
randomised and hand-built cases is the strongest pin available, because the reference has no
tests or golden vectors and cannot be built here (PARITY UNPINNED).

Python floats are IEEE f64 and `+ - * /` are correctly rounded, so results should agree with the
C oracle to the last bit wherever no libm function is involved.
"""
import math

from locityper_amd import cdefs

NOT_SAVED = 0xFFFFFFFF
NEG_INF = float("-inf")


# ---- seq/kmers.rs:163-202 ----
def kmers(seq: bytes, k: int, canonical=True, bits=128):
    mask = (1 << (2 * k)) - 1
    undef = (1 << bits) - 1
    rv_shift = 2 * k - 2 if canonical else 0
    fw = rv = 0
    k_1 = k - 1
    reset = k_1
    out = []
    enc_of = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3}
    for i, nt in enumerate(seq):
        enc = enc_of.get(nt)
        if enc is None:
            reset = i + k
            if i + 1 >= k:
                out.append(undef)
            continue
        fw = ((fw << 2) | enc) & mask
        if canonical:
            rv = (rv >> 2) | ((3 - enc) << rv_shift)
        if i >= reset:
            out.append(rv if (canonical and rv < fw) else fw)
        elif i + 1 >= k:
            out.append(undef)
    return out


# ---- seq/compl.rs:115-140 ----
def complexity_counts(seq: bytes, k: int, w: int):
    km = kmers(seq, k, canonical=False, bits=32)
    counts = {}
    unique = 0
    for x in km[:w - k + 1]:
        c = counts.get(x, 0)
        unique += c == 0
        counts[x] = c + 1
    res = [unique]
    for lag, new in zip(km, km[w - k + 1:]):
        if lag != new:
            c1 = counts.get(new, 0)
            unique += c1 == 0
            counts[new] = c1 + 1
            c2 = counts[lag]
            unique -= c2 == 1
            counts[lag] = c2 - 1
        res.append(unique)
    return res


class ParsingError(Exception):
    pass


class PyExplicitWeights:
    """ExplicitWeights of one haplotype (model/windows.rs:196-250): (value, running sum in units of 2^-32) per base pair."""
    SCALE = float(1 << 32)

    def __init__(self):
        self.weights = []
        self.sum = 0

    def extend_by(self, n, val):                        # 212-218
        i = int(val * self.SCALE)
        for _ in range(n):
            self.weights.append((val, self.sum))
            self.sum += i

    def finish(self):                                   # 221-224
        self.weights.append((self.weights[-1][0], self.sum))

    def at(self, i):
        return self.weights[i][0]

    def average(self, i, j):                            # 236-238: integer division, then to f64
        return float((self.weights[j][1] - self.weights[i][1]) // (j - i)) / self.SCALE


def load_explicit_weights(lines, lengths):
    """load_explicit_weights (windows.rs:257-317) on lines already split: (contig index or None if unknown, start, end, value)."""
    ws = [PyExplicitWeights() for _ in lengths]
    for contig, start, end, val in lines:
        if contig is None or contig >= len(lengths):
            continue                                    # unknown contig: ignored
        if end > lengths[contig] or start >= end:
            raise ValueError("interval out of range")   # InvalidInput (interv.rs:112-116)
        if val < 0.0 or val > 1.0:
            raise ParsingError("value must be in [0, 1]")
        if len(ws[contig].weights) != start:
            raise ParsingError("not fully covered")
        ws[contig].extend_by(end - start, val)
    for w, n in zip(ws, lengths):
        if not w.weights:
            raise ParsingError("haplotype missing")
        if len(w.weights) != n:
            raise ParsingError("not fully covered/has different length")
        w.finish()
    return ws


class PyLocus:
    """ContigSet + ContigInfos + UniqueKmers with the scalars of the path."""

    def __init__(self, alleles, counts, k, bg, params, insert_lnprob, insert_penalty, edit_thresholds):
        self.alleles = alleles                  # list[bytes]
        self.k = k
        self.bg, self.prm = bg, params
        self.insert_lnprob = insert_lnprob      # callable(sz) -> float (distribution math is tested separately)
        self.insert_penalty = insert_penalty
        self.edit_thresholds = edit_thresholds  # callable(read_len) -> (good, passable)
        neighb, ck = bg.neighb, params.complexity_k
        self.half_neighb = neighb // 2
        self.compl_mult = 1.0 / min(neighb + 1 - ck, 1 << (2 * ck))
        self.compl = [complexity_counts(a, ck, neighb) for a in alleles]       # windows.rs:404
        # UniqueKmers::new locs.rs:930-963
        self.unique = set()
        for a, cnt in zip(alleles, counts):
            km = kmers(a, k)
            assert len(km) == len(cnt)
            for x, c in zip(km, cnt):
                if c == 0:
                    self.unique.add(x)
        self.weight_mult = 1.0 / float(params.kmer_soft_thresh + 1 - params.kmer_hard_thresh)
        self.weight_interc = (1.0 - float(params.kmer_hard_thresh)) * self.weight_mult

    def neighb_complexity(self, contig, middle):       # windows.rs:447-452
        c = self.compl[contig]
        i = min(max(middle - self.half_neighb, 0), len(c) - 1)
        return c[i] * self.compl_mult

    explicit = None                                     # list[PyExplicitWeights] once --reg-weights are given

    def set_explicit_weights(self, lines):
        self.explicit = load_explicit_weights(lines, [len(a) for a in self.alleles])

    def window_explicit_weight(self, contig, i):        # windows.rs:409-413: mov_info[i].explicit_weight
        if self.explicit is None:
            return 1.0
        start = i + (self.bg.neighb - self.bg.window) // 2
        return self.explicit[contig].average(start, start + self.bg.window)

    def read_end_weight(self, contig, middle):          # windows.rs:493-503
        if middle is None or middle == cdefs.NONE_U32:
            return 0.0
        w = self.explicit[contig]
        n, u = len(w.weights), self.bg.window // 2
        return max(w.at(middle), w.at(max(middle - u, 0)), w.at(min(middle + u, n - 1)))

    def explicit_read_weight(self, pairs):              # windows.rs:683-693; pairs: [ln_prob, contig, ix1, mid1, ix2, mid2]
        if self.explicit is None:
            return 1.0
        s = 0.0
        for pa in pairs:
            s += max(self.read_end_weight(pa[1], pa[3]), self.read_end_weight(pa[1], pa[5]))
        return s / len(pairs)


def _decode_chunk(chunk):
    """ReadsChunk -> list of pairs: (seq1, seq2, [(contig, pos, flags, [(op, len)])])."""
    pairs = []
    for r in range(chunk.n_pairs):
        seqs = []
        for e in range(2):
            ln = int(chunk.mate_len[2 * r + e])
            off = int(chunk.mate_off[2 * r + e])
            s = bytearray()
            for i in range(ln):
                b = off + i
                if (int(chunk.nmask[b >> 5]) >> (b & 31)) & 1:
                    s.append(ord("N"))
                else:
                    s.append(b"ACGT"[(int(chunk.bases2[b >> 4]) >> (2 * (b & 15))) & 3])
            seqs.append(bytes(s))
        recs = []
        cbase = int(chunk.cigar_off[r])
        for i in range(int(chunk.aln_off[r]), int(chunk.aln_off[r + 1])):
            rec = chunk.recs[i]
            cig = [(int(w) & 15, int(w) >> 4) for w in
                   chunk.cigar[cbase + int(rec["cigar_rel"]):cbase + int(rec["cigar_rel"]) + int(rec["n_cigar"])]]
            recs.append((int(rec["contig"]), int(rec["pos"]), int(rec["flags"]), cig))
        pairs.append((seqs[0], seqs[1], recs))
    return pairs


class InvalidData(Exception):
    pass


def _make_aln(L, rec, rec_ix, read_end, primary):
    """Cigar::from_raw (+hard_to_soft for secondaries), Alignment::from_record, count_region_operations_fast,
    edit_distance, ErrorProfile::ln_prob (cigar.rs:295-320, aln.rs:147-157, 288-317; err_prof.rs:73-79, 212-221)."""
    contig, pos, flags, cig = rec
    if not cig:
        return None
    if primary and (cig[0][0] == cdefs.CIGAR_H or cig[-1][0] == cdefs.CIGAR_H):
        raise InvalidData("primary alignment has hard clipping")
    cig = list(cig)
    if cig[0][0] == cdefs.CIGAR_H:
        cig[0] = (cdefs.CIGAR_S, cig[0][1])
    if cig[-1][0] == cdefs.CIGAR_H:
        cig[-1] = (cdefs.CIGAR_S, cig[-1][1])
    m = x = ins = dl = ref_len = 0
    for op, ln in cig:
        if op == cdefs.CIGAR_EQ:
            m += ln; ref_len += ln
        elif op == cdefs.CIGAR_X:
            x += ln; ref_len += ln
        elif op == cdefs.CIGAR_D:
            dl += ln; ref_len += ln
        elif op == cdefs.CIGAR_I:
            ins += ln
        elif op == cdefs.CIGAR_S:
            pass
        else:
            raise InvalidData("unsupported CIGAR operation")
    if contig >= len(L.alleles):
        raise InvalidData("contig out of range")
    left = cig[0][1] if cig[0][0] == cdefs.CIGAR_S else 0
    right = cig[-1][1] if cig[-1][0] == cdefs.CIGAR_S else 0
    clen = len(L.alleles[contig])
    start, end = pos, pos + ref_len
    clip = min(left, start) + min(right, max(clen - end, 0))
    common = x + ins + clip
    lp = L.bg.op_lnprobs
    ln_prob = lp[0] * float(m) + lp[1] * float(x) + lp[2] * float(ins) + lp[3] * float(dl) + lp[4] * float(clip)
    return dict(contig=contig, start=start, end=end, rev=bool(flags & cdefs.FLAG_REVERSE), read_end=read_end,
                edit=common + dl, ln_prob=ln_prob, rec_ix=rec_ix)


class _Prelim:
    def __init__(self):
        self.alns = []
        self.pos = {}
        self.good = [NOT_SAVED, NOT_SAVED]
        self.passable = [NOT_SAVED, NOT_SAVED]
        self.best_edit = [NOT_SAVED, NOT_SAVED]
        self.best_lik = [NEG_INF, NEG_INF]

    def push(self, aln):      # locs.rs:298-344
        e = aln["read_end"]
        self.best_edit[e] = min(self.best_edit[e], aln["edit"])
        self.best_lik[e] = max(self.best_lik[e], aln["ln_prob"])
        new_ix = len(self.alns)
        save = aln["edit"] <= self.passable[e]
        if new_ix == 0 and not save:
            return False
        key = (e, aln["contig"], aln["start"] >> 7)
        ent = self.pos.get(key)
        if ent is not None:
            if save:
                if ent[0] == NOT_SAVED:
                    self.pos[key] = [new_ix, aln["start"]]
                    self.alns.append(aln)
                elif aln["ln_prob"] > self.alns[ent[0]]["ln_prob"]:
                    self.alns[ent[0]] = aln
                    ent[1] = aln["start"]
        else:
            if save:
                self.pos[key] = [new_ix, aln["start"]]
                self.alns.append(aln)
            else:
                self.pos[key] = [NOT_SAVED, aln["start"]]
        return save


def _is_secondary(flags):
    return bool(flags & (cdefs.FLAG_SECONDARY | cdefs.FLAG_SUPPL))


def _read_next_alns(L, seq_len, recs, ri, read_end, state, prelim):
    """locs.rs:502-567. Returns (well_mapped, next_record_index)."""
    if ri >= len(recs):
        raise InvalidData("no more records")
    rec = recs[ri]
    if seq_len == 0 or _is_secondary(rec[2]):
        raise InvalidData("bad primary")
    if rec[2] & cdefs.FLAG_UNMAPPED:
        return False, ri + 1
    aln = _make_aln(L, rec, ri, read_end, True)
    if aln is None:
        raise InvalidData("empty primary CIGAR")
    compl_v = L.neighb_complexity(aln["contig"], (aln["start"] + aln["end"]) // 2) \
        if L.bg.technology == cdefs.TECH_ILLUMINA else 1.0
    good, passable = L.edit_thresholds(seq_len)
    thr = good
    if compl_v <= L.prm.poor_compl:
        thr = max(good, int(L.prm.poor_compl_edit * float(seq_len)))
        passable += thr - good
    prelim.good[read_end], prelim.passable[read_end] = thr, passable
    if not prelim.push(aln):
        ri += 1
        while ri < len(recs) and _is_secondary(recs[ri][2]):
            ri += 1
        return False, ri
    ri += 1
    while ri < len(recs) and _is_secondary(recs[ri][2]):
        a = _make_aln(L, recs[ri], ri, read_end, False)
        if a is not None:
            prelim.push(a)
        ri += 1
    best = prelim.best_edit[read_end]
    req = passable if L.prm.strict_subset else thr
    if best > req:
        return False, ri
    state["weight"] *= 1.0 if best <= good else math.sqrt(float(good) / float(best))
    return True, ri


def _contig_pairs(L, alns, i, j, k, max_alns, unm_ins_pen):      # locs.rs:746-799
    out = []
    buf = [NEG_INF] * (k - j)
    for ix1 in range(i, j):
        a1 = alns[ix1]
        max1 = NEG_INF
        for ix2 in range(j, k):
            a2 = alns[ix2]
            if a1["rev"] != a2["rev"]:
                ins = max(a1["end"], a2["end"]) - min(a1["start"], a2["start"])
                prob = a1["ln_prob"] + a2["ln_prob"] + L.insert_lnprob(ins)
                if math.isfinite(prob):
                    max1 = max(max1, prob)
                    buf[ix2 - j] = max(buf[ix2 - j], prob)
                    out.append([prob, a1["contig"], a1["rec_ix"], (a1["start"] + a1["end"]) // 2,
                                a2["rec_ix"], (a2["start"] + a2["end"]) // 2])
        alone = a1["ln_prob"] + unm_ins_pen
        if alone >= max1:
            out.append([alone, a1["contig"], a1["rec_ix"], (a1["start"] + a1["end"]) // 2, cdefs.NONE_U32, cdefs.NONE_U32])
    for ix2 in range(j, k):
        a2 = alns[ix2]
        alone = a2["ln_prob"] + unm_ins_pen
        if alone >= buf[ix2 - j]:
            out.append([alone, a2["contig"], cdefs.NONE_U32, cdefs.NONE_U32, a2["rec_ix"], (a2["start"] + a2["end"]) // 2])
    order = sorted(range(len(out)), key=lambda t: (-out[t][0], t))      # stable: ties keep push order
    out = [out[t] for t in order]
    thresh = out[0][0] - L.prm.prob_diff
    keep = 0
    while keep < min(len(out), max_alns) and out[keep][0] >= thresh:
        keep += 1
    return out[:keep]


def load(L, chunk):
    """AllAlignments::load + recover_and_group_alignments without hap alns (locs.rs:1085-1185, 1237-1288).

    Returns per pair: status, weight, unmapped_prob, (uk1, uk2), [pair alns as
    (ln_prob, contig, ix1, mid1, ix2, mid2)]."""
    res = []
    paired = bool(L.bg.is_paired)
    boundary = L.prm.boundary_size - L.prm.tweak
    for seq1, seq2, recs in _decode_chunk(chunk):
        state = {"weight": 1.0}
        prelim = _Prelim()
        ok, ri = _read_next_alns(L, len(seq1), recs, 0, 0, state, prelim)
        if paired and ok:
            ok, ri = _read_next_alns(L, len(seq2), recs, ri, 1, state, prelim)
        if not ok:
            res.append((cdefs.READ_POORLY_MAPPED, 0.0, 0.0, (0, 0), []))
            continue
        if not any(boundary <= (a["start"] + a["end"]) // 2 < len(L.alleles[a["contig"]]) - boundary
                   for a in prelim.alns):
            res.append((cdefs.READ_OUT_OF_BOUNDS, 0.0, 0.0, (0, 0), []))
            continue
        uks = []
        for s in ((seq1, seq2) if paired else (seq1,)):
            if not s:
                uks.append(0)
                continue
            km = kmers(s, L.k)
            cnt, i = 0, 0
            while i < len(km):
                x = km[i]
                i += 1
                if x in L.unique:
                    cnt = min(cnt + 1, 0xFFFF)
                    i += L.k - 1
            uks.append(cnt)
        while len(uks) < 2:
            uks.append(0)
        w = L.weight_interc + float((uks[0] + uks[1]) & 0xFFFF) * L.weight_mult
        state["weight"] *= min(max(w, 0.0), 1.0)
        weight = state["weight"]
        if not (prelim.best_edit[0] <= prelim.good[0] and prelim.best_edit[1] <= prelim.good[1]):
            res.append((cdefs.READ_POORLY_MAPPED, 0.0, 0.0, (0, 0), []))
            continue
        for a in prelim.alns:
            a["ln_prob"] = a["ln_prob"] - prelim.best_lik[a["read_end"]]
        max_alns = 10 if weight >= L.prm.min_weight else 2
        pairs = []
        if paired:
            pen = L.insert_penalty
            unm_ins = L.prm.unmapped_penalty + pen
            # pop order: contig asc, end asc, ln_prob desc (ties: input order)
            tmp = sorted(prelim.alns, key=lambda a: (a["contig"], a["read_end"], -a["ln_prob"], a["rec_ix"]))
            kept = []
            cur, i, j = 0, 0, None
            for a in tmp:
                k = len(kept)
                if cur != a["contig"]:
                    if i < k:
                        pairs += _contig_pairs(L, kept, i, k if j is None else min(j, k), k, max_alns, unm_ins)
                    cur, i, j = a["contig"], k, None
                if a["read_end"] == 0:
                    if k - i < max_alns:
                        kept.append(a)
                else:
                    j = k if j is None else min(j, k)
                    if k - j < max_alns:
                        kept.append(a)
            k = len(kept)
            if i < k:
                pairs += _contig_pairs(L, kept, i, k if j is None else min(j, k), k, max_alns, unm_ins)
            weight = weight * L.explicit_read_weight(pairs)            # locs.rs:860
            unm = weight * (2.0 * L.prm.unmapped_penalty + pen)
        else:
            tmp = sorted(prelim.alns, key=lambda a: (a["contig"], -a["ln_prob"], a["rec_ix"]))
            cur, thresh, saved = None, float("nan"), 0
            for a in tmp:
                if cur != a["contig"]:
                    cur, thresh, saved = a["contig"], a["ln_prob"] - L.prm.prob_diff, 0
                if a["ln_prob"] >= thresh and saved < max_alns:
                    pairs.append([a["ln_prob"], a["contig"], a["rec_ix"], (a["start"] + a["end"]) // 2,
                                  cdefs.NONE_U32, cdefs.NONE_U32])
                    saved += 1
            weight = weight * L.explicit_read_weight(pairs)            # locs.rs:903
            unm = weight * L.prm.unmapped_penalty
        for pa in pairs:
            pa[0] = pa[0] * weight
        status = cdefs.READ_GOOD if weight >= L.prm.min_weight else cdefs.READ_FEW_KMERS
        res.append((status, weight, unm, (uks[0], uks[1]), [tuple(x) for x in pairs]))
    return res


def run_filter(matrix, genotypes, priors=None):       # solve.rs:101-119
    scores = []
    n_reads = len(matrix[0]) if len(matrix) else 0
    for g, ids in enumerate(genotypes):
        best = list(matrix[ids[0]])
        for a in ids[1:]:
            row = matrix[a]
            best = [max(b, v) for b, v in zip(best, row)]
        s = -0.0
        for v in best:
            s += v
        scores.append((priors[g] if priors is not None else 0.0) + s)
    assert n_reads >= 0
    return scores


def truncate_ixs(scores, ixs, filt_diff, min_size, threads):      # solve.rs:52-84
    ixs = sorted(ixs, key=lambda i: (-scores[i], i))
    n = len(ixs)
    best, worst = scores[ixs[0]], scores[ixs[-1]]
    thresh = best - filt_diff
    if min_size >= n or worst >= thresh:
        return ixs

    def ppoint(t):
        m = 0
        while m < n and scores[ixs[m]] >= t:
            m += 1
        return m
    m = ppoint(thresh)
    if m < min_size:
        thresh = scores[ixs[min_size - 1]]
        m = ppoint(thresh)
    m = min(max(m, threads), n)
    return ixs[:m]


def gen_combinations_with_repl(n, size):      # ext/vec.rs:298-339
    out = []

    def rec(buf, start, depth):
        if depth + 1 == size:
            for el in range(start, n):
                buf[depth] = el
                out.append(tuple(buf))
        else:
            for el in range(start, n):
                buf[depth] = el
                rec(buf, el, depth + 1)
    if n and size:
        rec([0] * size, 0, 0)
    return out


# ---------------------------------------------------------------- solver stages (second transliteration)
def extend_read_gt_alns(pair_alns, unmapped_prob, ids, prob_diff):
    """GenotypeWindows::extend_read_gt_alns (windows.rs:762-797) for one read.

    pair_alns: [(ln_prob, contig, ix1, mid1, ix2, mid2)] contig-ascending, ln_prob-descending.
    Returns [(ln_prob, contig_ix or 0xFF, mid1, mid2)] sorted, thresholded."""
    thresh = unmapped_prob - prob_diff
    out = []
    for i, cid in enumerate(ids):
        alns = [pa for pa in pair_alns if pa[1] == cid]
        if alns:
            thresh = max(thresh, alns[0][0] - prob_diff)
            for pa in alns:
                if pa[0] >= thresh:
                    out.append((pa[0], i, pa[3], pa[5]))
                else:
                    break
    if unmapped_prob >= thresh:
        out.append((unmapped_prob, 0xFF, cdefs.NONE_U32, cdefs.NONE_U32))
    order = sorted(range(len(out)), key=lambda t: (-out[t][0], t))
    out = [out[t] for t in order]
    keep = 0
    while keep < len(out) and out[keep][0] >= thresh:
        keep += 1
    return out[:keep]


def window_ix(reg_start, n_windows, window, shift, middle):
    """get_shifted_window_ix (windows.rs:465-470) + middle_window (62-68)."""
    if middle == cdefs.NONE_U32:
        return 0
    if reg_start <= middle < reg_start + n_windows * window:
        return (middle - reg_start) // window + shift
    return 1


def depth_lik_diff_counts(w1, w2, w3, w4):
    """The (window -> depth change) bookkeeping of ReadAssignment::depth_lik_diff (assgn.rs:259-284)."""
    c1 = -1
    if w2 == w1:
        c1 -= 1; c2 = 0
    else:
        c2 = -1
    if w3 == w1:
        c1 += 1; c3 = 0
    elif w3 == w2:
        c2 += 1; c3 = 0
    else:
        c3 = 1
    if w4 == w1:
        c1 += 1; c4 = 0
    elif w4 == w2:
        c2 += 1; c4 = 0
    elif w4 == w3:
        c3 += 1; c4 = 0
    else:
        c4 = 1
    assert c1 + c2 + c3 + c4 == 0
    return [(w1, c1), (w2, c2), (w3, c3), (w4, c4)]
