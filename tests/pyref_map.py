"""Python restatement of the candidate-generation slice (locityper_amd/csrc/lcty_map.hip): TEST INFRASTRUCTURE, the checker the HIP
kernel must equal bit for bit. The algorithm has no counterpart in the reference tree (the reference calls an external mapper,
src/command/genotype.rs:962-1094); it is stated in lcty_map.hip and include/locityper_hip.h and restated here independently, on
plain Python integers and lists.
"""
import numpy as np

FLAG_UNMAPPED, FLAG_REVERSE, FLAG_MATE2, FLAG_SECONDARY = 0x4, 0x10, 0x80, 0x100
ENC = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3}
MAX_HITS, PER_SEED = 1024, 64


def build_index(seqs, seq_off, basis, k):
    """canonical k-mer -> [(basis index, position, forward-is-canonical)] in (allele, position) order"""
    index = {}
    for b, a in enumerate(basis):
        s = bytes(seqs[int(seq_off[a]):int(seq_off[a + 1])])
        for i in range(len(s) - k + 1):
            w = s[i:i + k]
            if any(c not in ENC for c in w):
                continue
            fw = 0
            for c in w:
                fw = (fw << 2) | ENC[c]
            rv = 0
            for c in reversed(w):
                rv = (rv << 2) | (3 - ENC[c])
            fwd = fw <= rv
            index.setdefault(fw if fwd else rv, []).append((b, i, fwd))
    return index


def mate_bases(ch, m):
    off, ln = int(ch.mate_off[m]), int(ch.mate_len[m])
    bases = [(int(ch.bases2[(off + i) >> 4]) >> (2 * ((off + i) & 15))) & 3 for i in range(ln)]
    isn = [bool((int(ch.nmask[(off + i) >> 5]) >> ((off + i) & 31)) & 1) for i in range(ln)]
    return bases, isn


NEG = -(1 << 29)


def gapped(L, diag, ref, equal_ref, p):
    """The clipped candidate again, with gaps: gap-affine in a band of +-p.band diagonals around `diag`, starting and ending on an
    aligned base. -> (score, first read position, last read position, reference position of the first column, CIGAR words
    without the soft clips) or None. equal_ref(i, refpos): base i of the read end (alignment orientation) equals ref[refpos]."""
    B = p.band
    W = 2 * B + 1
    M, E, F = [NEG] * W, [NEG] * W, [NEG] * W
    dirs = [[0] * W for _ in range(L)]
    best_total, end_i, end_k = None, 0, 0
    for i in range(L):
        left_m = left_e = NEG
        fresh = p.end_bonus if i == 0 else 0
        for k in range(W):
            refpos = diag + i + (k - B)
            inref = 0 <= refpos < len(ref)
            om, oe, of = M[k], E[k], F[k]
            rm, rf = (M[k + 1], F[k + 1]) if k + 1 < W else (NEG, NEG)
            prev, code = om, 1
            if oe > prev:
                prev, code = oe, 2
            if of > prev:
                prev, code = of, 3
            if fresh > prev:
                prev, code = fresh, 0
            nm = prev + (p.match if equal_ref(i, refpos) else -p.mismatch) if inref else NEG
            fo, fe = rm - p.gap_open, rf - p.gap_extend
            nf, fcode = (fe, 1) if fe > fo else (fo, 0)
            eo, ee = left_m - p.gap_open, left_e - p.gap_extend
            ne, ecode = (ee, 1) if ee > eo else (eo, 0)
            if not inref:
                ne = NEG
            nm = NEG if nm < NEG // 2 else nm
            nf = NEG if nf < NEG // 2 else nf
            ne = NEG if ne < NEG // 2 else ne
            dirs[i][k] = code | (ecode << 2) | (fcode << 3)
            M[k], E[k], F[k] = nm, ne, nf
            left_m, left_e = nm, ne
            if nm > NEG:
                total = nm + (p.end_bonus if i + 1 == L else 0)
                if best_total is None or total > best_total:
                    best_total, end_i, end_k = total, i, k
    if best_total is None:
        return None
    i, k, state, ops = end_i, end_k, 0, []                      # ops last first, as (op, length) runs
    def emit(op):
        if ops and ops[-1][0] == op:
            ops[-1][1] += 1
        else:
            ops.append([op, 1])
    while True:
        d = dirs[i][k]
        if state == 0:
            emit(7 if equal_ref(i, diag + i + (k - B)) else 8)
            c = d & 3
            if c == 0:
                break
            state = c - 1
            i -= 1
        elif state == 1:
            emit(2)
            state = 1 if (d >> 2) & 1 else 0
            k -= 1
        else:
            emit(1)
            state = 2 if (d >> 3) & 1 else 0
            i -= 1
            k += 1
    words = [(n << 4) | op for op, n in reversed(ops)]
    return best_total, i, end_i, diag + i + (k - B), words


def map_mate(bases, isn, index, seqs, seq_off, basis, p):
    """-> list of (allele, strand, pos, score, cigar words), primary first; [] = unmapped"""
    L, k = len(bases), p.k
    hits = []
    if L >= k:
        span = L - k
        starts = [i * p.stride for i in range(span // p.stride + 1)] + ([span] if span % p.stride else [])
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
            if len(places) > (p.max_occ or 4 * len(basis)):
                continue
            for (b, pa, allele_fwd) in places[:PER_SEED]:
                strand = 0 if read_fwd == allele_fwd else 1
                diag = pa - (L - k - pr) if strand else pa - pr
                hits.append((b * 2 + strand, diag))
        hits = hits[:MAX_HITS]
    votes = {}
    for g, d in hits:
        votes.setdefault(g, {}).setdefault(d, 0)
        votes[g][d] += 1
    if not votes:
        return []
    group_best = {g: max(dd.values()) for g, dd in votes.items()}
    max_votes = max(group_best.values())
    # candidate diagonals in (allele, strand, diagonal) order, the first 64
    diags = [(g, d) for g in sorted(votes) for d in sorted(votes[g])
             if votes[g][d] >= max(p.min_votes, 1) and 2 * votes[g][d] >= group_best[g] and 2 * group_best[g] >= max_votes][:64]
    per_group = {}
    for g, diag in diags:
        strand, allele = g & 1, int(basis[g >> 1])
        ref = bytes(seqs[int(seq_off[allele]):int(seq_off[allele + 1])])

        def equal(i):
            src = L - 1 - i if strand else i
            if isn[src]:
                return False
            e = 3 - bases[src] if strand else bases[src]
            return ENC.get(ref[diag + i], 4) == e
        i_lo, i_hi = max(0, -diag), max(0, min(L, len(ref) - diag))
        score, s_best, e_best = None, 0, 0
        run, run_s, opened = 0, 0, False
        for i in range(i_lo, i_hi):
            fresh = p.end_bonus if i == 0 else 0
            if not opened or fresh > run:
                run, run_s, opened = fresh, i, True
            run += p.match if equal(i) else -p.mismatch
            total = run + (p.end_bonus if i + 1 == L else 0)
            if score is None or total > score:
                score, s_best, e_best = total, run_s, i + 1
        if score is None:
            continue
        if g in per_group and per_group[g][4] >= score:          # the best score of the (allele, strand); the smallest diagonal on ties
            continue
        cig = []
        if s_best > 0:
            cig.append((s_best << 4) | 4)
        prev, ln = None, 0
        for i in range(s_best, e_best):
            eq = equal(i)
            if prev is not None and eq != prev:
                cig.append((ln << 4) | (7 if prev else 8))
                ln = 0
            prev, ln = eq, ln + 1
        cig.append((ln << 4) | (7 if prev else 8))
        if e_best < L:
            cig.append(((L - e_best) << 4) | 4)
        per_group[g] = (g, allele, strand, diag + s_best, score, cig, diag, s_best, e_best)
    cands = []
    for g in sorted(per_group):
        (_, allele, strand, pos, score, cig, diag, s_best, e_best) = per_group[g]
        if getattr(p, "band", 0) > 0 and (s_best > 0 or e_best < L):
            ref = bytes(seqs[int(seq_off[allele]):int(seq_off[allele + 1])])

            def equal_ref(i, refpos, strand=strand, ref=ref):
                src = L - 1 - i if strand else i
                if isn[src]:
                    return False
                e = 3 - bases[src] if strand else bases[src]
                return ENC.get(ref[refpos], 4) == e
            got = gapped(L, diag, ref, equal_ref, p)
            if got is not None and got[0] > score:
                gscore, first, last, gpos, words = got
                cig = ([(first << 4) | 4] if first > 0 else []) + words + ([((L - 1 - last) << 4) | 4] if last < L - 1 else [])
                pos, score = gpos, gscore
        cands.append((g, allele, strand, pos, score, cig))
    if not cands:
        return []
    top = max(c[4] for c in cands)
    gp = min(c[0] for c in cands if c[4] == top)
    kept = [c for c in cands if c[0] == gp] + [c for c in cands if c[0] != gp and c[4] >= p.min_score]
    return [(a, s, pos, sc, cig) for (_, a, s, pos, sc, cig) in kept]


def map_chunk(ch, seqs, seq_off, basis, p, paired=True):
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
            found = map_mate(bases, isn, index, seqs, seq_off, basis, p)
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
