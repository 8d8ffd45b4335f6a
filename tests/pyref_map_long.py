"""Python restatement of the LONG route of candidate generation (locityper_amd/csrc/lcty_map_long.hip): TEST INFRASTRUCTURE, the
checker the HIP kernels must equal bit for bit. Read ends of any length on up to 256 basis alleles: seeds -> anchors -> one chain per
(basis allele, strand) -> gap-affine alignment along the chain (banded pieces between the anchors, banded extensions at the two ends).
The reference leaves this to minimap2 (src/command/genotype.rs:990-1002); the algorithm is this build's own, stated in
lcty_map_long.hip and include/locityper_hip.h and restated here independently on plain Python integers and lists.

Coordinates of an (allele, strand) group: q = position on the read end AS SEQUENCED, t = position on the allele in the read's
orientation (strand 0: the allele itself; strand 1: its reverse complement). Every group is handled in these coordinates; the
record is turned into BAM orientation at the very end.
"""
import numpy as np

from tests.pyref_map import ENC, FLAG_MATE2, FLAG_REVERSE, FLAG_SECONDARY, FLAG_UNMAPPED, build_index, mate_bases

NEG = -(1 << 29)


def skew_cost(d):
    return 0 if d == 0 else 2 + d


def seed_starts(L, k, stride):
    if L < k:
        return []
    span = L - k
    return [i * stride for i in range(span // stride + 1)] + ([span] if span % stride else [])


def chains(bases, isn, index, basis_len, p):
    """-> {g: (anchors [(q, t, f, back, cnt)], index of the best chain end)}"""
    L, k = len(bases), p.k
    starts = seed_starts(L, k, p.stride)
    cap = 2 * len(starts)
    max_occ = p.max_occ or 4 * len(basis_len)
    groups, best = {}, {}
    for pr in starts:
        if any(isn[pr:pr + k]):
            continue
        fw = 0
        for e in bases[pr:pr + k]:
            fw = (fw << 2) | e
        rv = 0
        for e in reversed(bases[pr:pr + k]):
            rv = (rv << 2) | (3 - e)
        read_fwd = fw <= rv
        places = index.get(fw if read_fwd else rv, [])
        if len(places) > max_occ:
            continue
        for (b, pa, allele_fwd) in places:
            strand = 0 if read_fwd == allele_fwd else 1
            g = 2 * b + strand
            q, t = pr, (pa if strand == 0 else basis_len[b] - k - pa)
            lst = groups.setdefault(g, [])
            if len(lst) >= cap:
                continue
            f, back, cnt = k, 0, 1
            for o in range(1, min(p.chain_back, len(lst)) + 1):             # the most recent anchor of the group first
                qj, tj, fj, _, cj = lst[-o]
                dq, dt = q - qj, t - tj
                if dq <= 0 or dt <= 0 or dq > p.chain_gap or dt > p.chain_gap:
                    continue
                sk = abs(dq - dt)
                if sk > p.chain_skew or (sk and (dq < k or dt < k)):
                    continue
                v = fj + min(dq, dt, k) - skew_cost(sk)
                if v > f:
                    f, back, cnt = v, o, cj + 1
            lst.append((q, t, f, back, cnt))
            if g not in best or f > best[g][0]:
                best[g] = (f, len(lst) - 1)
    return {g: (groups[g], best[g][1]) for g in best}


class Emit:
    """CIGAR runs in the order they are met (right to left)"""
    def __init__(self):
        self.runs = []

    def __call__(self, op, n=1):
        if n == 0:
            return
        if self.runs and self.runs[-1][0] == op:
            self.runs[-1][1] += n
        else:
            self.runs.append([op, n])


def segment(q0, n, t0, m, free_start, free_end, eq, p, emit):
    """Gap-affine alignment of read bases [q0, q0 + n) with allele bases [t0, t0 + m) over the nodes (i, j) = (read bases taken, allele
    bases taken) of a band of diagonals; H = best of (base step, deletion, insertion), gaps open from H.
      fixed start, fixed end (a piece between two anchors): from node (0, 0) to node (n, m); band min(0, m - n) - B .. max(0, m - n) + B
      fixed start, free end (right extension): ends on the base step with the best total (end bonus at i = n), or not at all;
                                               band -B .. +B, m = what is left of the allele
      free start, fixed end (left extension): a base step may start afresh (end bonus from i = 0); band (m - n) - B .. (m - n) + B
    -> (score, read bases left unaligned at the free side, allele bases taken). The runs go to `emit` right to left."""
    B = p.band
    if free_start:
        dlo, dhi = (m - n) - B, (m - n) + B
    elif free_end:
        dlo, dhi = -B, B
    else:
        dlo, dhi = min(0, m - n) - B, max(0, m - n) + B
    W = dhi - dlo + 1
    H, F = [NEG] * W, [NEG] * W
    dirs = [[0] * W for _ in range(n + 1)]
    best = (p.end_bonus if n == 0 else 0, 0, -dlo)                          # free end: no extension at all
    for i in range(n + 1):
        e = hleft = NEG
        for kk in range(W):
            j = i + dlo + kk
            if j < 0 or j > m:
                H[kk] = F[kk] = NEG
                e = hleft = NEG
                continue
            mc, code = NEG, 0
            if i >= 1 and j >= 1:
                hd = H[kk]
                fr = (p.end_bonus if i == 1 else 0) if free_start else NEG
                base = hd
                if fr > hd:
                    base, code = fr, 3
                if base > NEG // 2:
                    mc = base + (p.match if eq(q0 + i - 1, t0 + j - 1) else -p.mismatch)
            if i == 0 and j == 0 and not free_start:
                mc = 0
            f, fbit = NEG, 0
            if i >= 1:
                up_h, up_f = (H[kk + 1], F[kk + 1]) if kk + 1 < W else (NEG, NEG)
                fo, fe = up_h - p.gap_open, up_f - p.gap_extend
                f, fbit = (fe, 1) if fe > fo else (fo, 0)
            ebit = 0
            if j >= 1:
                eo, ee = hleft - p.gap_open, e - p.gap_extend
                e, ebit = (ee, 1) if ee > eo else (eo, 0)
            else:
                e = NEG
            if mc < NEG // 2:
                mc = NEG
            if f < NEG // 2:
                f = NEG
            if e < NEG // 2:
                e = NEG
            h = mc
            if e > h:
                h, code = e, 1
            if f > h:
                h, code = f, 2
            H[kk], F[kk] = h, f
            dirs[i][kk] = code | (ebit << 2) | (fbit << 3)
            hleft = h
            if free_end and mc > NEG and i >= 1:
                total = mc + (p.end_bonus if i == n else 0)
                if total > best[0]:
                    best = (total, i, kk)
    if free_end:
        score, i, kk = best
        state = "M"
    else:
        i, kk = n, m - n - dlo
        score = H[kk]
        fresh_all = (p.end_bonus if n == 0 else 0) if free_start else NEG
        if free_start and not score > fresh_all:
            return fresh_all, n, 0                                          # nothing left of the first anchor is aligned
        assert score > NEG
        state = "H"
    left, j_end = n - i if free_end else 0, i + dlo + kk
    while True:
        j = i + dlo + kk
        d = dirs[i][kk]
        if state == "H":
            if not free_start and i == 0 and j == 0:
                break
            c = d & 3
            state = "M" if c in (0, 3) else ("E" if c == 1 else "F")
        if state == "M":
            if not free_start and i == 0 and j == 0:
                break
            emit(7 if eq(q0 + i - 1, t0 + j - 1) else 8)
            i -= 1
            if (d & 3) == 3:
                return score, i, m - (i + dlo + kk)                        # started afresh: i read bases stay clipped
            state = "H"
        elif state == "E":
            emit(2)
            kk -= 1
            state = "E" if (d >> 2) & 1 else "H"
        else:
            emit(1)
            i -= 1
            kk += 1
            state = "F" if (d >> 3) & 1 else "H"
    if free_end:
        return score, left, j_end
    return score, 0, m


def align_chain(anchors, end, L, alen, eq, p):
    """-> (score, first allele position t, one past the last, CIGAR runs LEFT TO RIGHT in (q, t) coordinates incl. soft clips)"""
    k = p.k
    emit = Emit()
    q, t = anchors[end][0], anchors[end][1]
    n = L - (q + k)
    sc, clipped, t_taken = segment(q + k, n, t + k, alen - (t + k), False, True, eq, p, emit)
    # `emit` holds the extension right to left; the clip goes to its right
    runs_ext = emit.runs
    emit = Emit()
    emit(4, clipped)
    for op, ln in runs_ext:
        emit(op, ln)
    t_end = t + k + t_taken
    score = sc + k * p.match
    emit(7, k)
    cur_q, cur_t = q, t
    at = end
    while anchors[at][3]:
        at -= anchors[at][3]
        q, t = anchors[at][0], anchors[at][1]
        if q + k > cur_q or t + k > cur_t:                                   # overlapping seeds of one diagonal
            assert cur_q - q == cur_t - t
            emit(7, cur_q - q)
            score += (cur_q - q) * p.match
        else:
            sc, _, _ = segment(q + k, cur_q - (q + k), t + k, cur_t - (t + k), False, False, eq, p, emit)
            score += sc + k * p.match
            emit(7, k)
        cur_q, cur_t = q, t
    sc, lead, t_taken = segment(0, cur_q, 0, cur_t, True, False, eq, p, emit)
    score += sc
    emit(4, lead)
    return score, cur_t - t_taken, t_end, [(op, ln) for op, ln in reversed(emit.runs)]


def map_mate_long(bases, isn, index, seqs, seq_off, basis, p):
    """-> list of (allele, strand, pos, score, cigar words), primary first; [] = unmapped"""
    L, k = len(bases), p.k
    basis_len = [int(seq_off[a + 1]) - int(seq_off[a]) for a in basis]
    ch = chains(bases, isn, index, basis_len, p)
    if not ch:
        return []
    top = max(lst[e][2] for lst, e in ch.values())
    comp = {0: 3, 1: 2, 2: 1, 3: 0}
    cands = []
    for g in sorted(ch):
        lst, e = ch[g]
        if 2 * lst[e][2] < top or lst[e][4] < max(p.min_votes, 1):
            continue
        strand, allele = g & 1, int(basis[g >> 1])
        ref = bytes(seqs[int(seq_off[allele]):int(seq_off[allele + 1])])
        alen = len(ref)

        def eq(q, t, strand=strand, ref=ref, alen=alen):
            if isn[q]:
                return False
            r = ENC.get(ref[alen - 1 - t] if strand else ref[t], 4)
            if r == 4:
                return False
            return bases[q] == (comp[r] if strand else r)
        score, t0, t1, runs = align_chain(lst, e, L, alen, eq, p)
        if strand:
            runs = list(reversed(runs))
            pos = alen - t1
        else:
            pos = t0
        cands.append((g, allele, strand, pos, score, [(ln << 4) | op for op, ln in runs]))
    if not cands:
        return []
    top = max(c[4] for c in cands)
    gp = min(c[0] for c in cands if c[4] == top)
    kept = [c for c in cands if c[0] == gp] + [c for c in cands if c[0] != gp and c[4] >= p.min_score]
    return [(a, s, pos, sc, cig) for (_, a, s, pos, sc, cig) in kept]


def map_chunk_long(ch, seqs, seq_off, basis, p, paired=True):
    """-> (aln_off, records [(pos, contig, flags, n_cigar, cigar_rel)], cigar_off, cigar words, primary strands per mate)"""
    index = build_index(seqs, seq_off, basis, p.k)
    aln_off, cig_off, recs, cigar, strands = [0], [0], [], [], []
    for pair in range(ch.n_pairs):
        pair_cig = len(cigar)
        for e in range(2):
            m = 2 * pair + e
            if int(ch.mate_len[m]) == 0:
                strands.append(0)
                continue
            bases, isn = mate_bases(ch, m)
            found = map_mate_long(bases, isn, index, seqs, seq_off, basis, p)
            mate2 = FLAG_MATE2 if (paired and e == 1) else 0
            if not found:
                recs.append((0, 0, FLAG_UNMAPPED | mate2, 0, len(cigar) - pair_cig))
                strands.append(0)
                continue
            strands.append(found[0][1])
            for j, (a, s, pos, sc, cig) in enumerate(found):
                flags = (FLAG_REVERSE if s else 0) | (FLAG_SECONDARY if j else 0) | mate2
                recs.append((pos, a, flags, len(cig), len(cigar) - pair_cig))
                cigar.extend(cig)
        aln_off.append(len(recs))
        cig_off.append(len(cigar))
    return (np.array(aln_off, dtype=np.uint64), recs, np.array(cig_off, dtype=np.uint64), np.array(cigar, dtype=np.uint32), strands)
