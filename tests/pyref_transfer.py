"""Second, independent transliteration of the alignment-recovery code (TEST INFRASTRUCTURE): written line by line from the Rust
(seq/cigar.rs:969-1466, seq/wfa.rs:49-84 and 162-365, seq/transfer.rs:90-124), separately from oracle/lcty_oracle_transfer.c.
The WFA2-lib calls are the one thing taken from the oracle (`dp_align`: exact gap-affine optimum with the tie rule documented
there); everything around them — CigarIndex, the two-CIGAR walk, anchors, clipping, optimize — is restated here on its own, so that
`tests/test_oracle_transfer.py` can hold the C restatement against it CIGAR for CIGAR."""
from tests import oracle_ffi as O

SPARSE_STEP_PWR = 8
SPARSE_MASK = (1 << SPARSE_STEP_PWR) - 1
MISMATCH, GAP_OPEN, GAP_EXTEND = 4, 6, 1                     # Penalties::default, wfa.rs:30-38
SAFE_MISMATCH = (2 * GAP_OPEN + 2 * GAP_EXTEND) // MISMATCH  # wfa.rs:212
DROPPED = -(1 << 30)


def consumes(op):
    """(query, ref) — cigar.rs:131-145"""
    return {"=": (True, True), "X": (True, True), "M": (True, True), "I": (True, False), "S": (True, False),
            "D": (False, True), "H": (False, False)}[op]


def invert_op(op):
    return {"I": "D", "S": "D", "D": "I"}.get(op, op)        # cigar.rs:147-159


class Cig:
    def __init__(self, tuples=None):
        self.t, self.qlen, self.rlen = [], 0, 0
        for op, ln in tuples or []:
            self.push_unchecked(op, ln)

    @staticmethod
    def parse(s):
        out, num = [], ""
        for ch in s:
            if ch.isdigit(): num += ch
            else: out.append((ch, int(num))); num = ""
        return Cig(out)

    def push_unchecked(self, op, ln):                        # cigar.rs:343-352
        q, r = consumes(op)
        self.qlen += ln if q else 0; self.rlen += ln if r else 0
        self.t.append([op, ln])

    def push_checked(self, op, ln):                          # cigar.rs:355-363
        q, r = consumes(op)
        self.qlen += ln if q else 0; self.rlen += ln if r else 0
        if self.t and self.t[-1][0] == op: self.t[-1][1] += ln
        else: self.t.append([op, ln])

    def clear(self):
        self.t, self.qlen, self.rlen = [], 0, 0

    def hard_to_soft(self):                                  # cigar.rs:309-320
        if not self.t: return
        if self.t[0][0] == "H": self.t[0][0] = "S"; self.qlen += self.t[0][1]
        if self.t[-1][0] == "H": self.t[-1][0] = "S"; self.qlen += self.t[-1][1]

    def __str__(self):
        return "".join(f"{n}{op}" for op, n in self.t)


class CigarIndex:                                            # cigar.rs:1087-1162
    def __init__(self, cig):
        self.positions, qpos, rpos = [], 0, 0
        qr, rq = [], []

        def update(index, cigar_ix, ln, pos1, pos2, consumes_other):        # update_sparse_index, 973-986
            for i in range(len(index), ((pos1 + ln - 1) >> SPARSE_STEP_PWR) + 1):
                sparse_pos1 = i << SPARSE_STEP_PWR
                index.append((cigar_ix, pos2 + (sparse_pos1 - pos1 if consumes_other else 0)))
        for ix, (op, ln) in enumerate(cig.t):
            self.positions.append((qpos, rpos))
            cq, cr = consumes(op)
            old_qpos = qpos
            if cq:
                update(qr, ix, ln, qpos, rpos, cr); qpos += ln
            if cr:
                update(rq, ix, ln, rpos, old_qpos, cq); rpos += ln
        qr.append((len(cig.t) - 1, cig.rlen)); rq.append((len(cig.t) - 1, cig.qlen))
        self.sparse = [qr, rq]

    def find_approx_position(self, qpos, ref_to_query):
        idx = self.sparse[1 if ref_to_query else 0]
        i = qpos >> SPARSE_STEP_PWR
        (min_ix, rpos1), (max_ix, rpos2) = idx[i], idx[i + 1]
        return min_ix, max_ix, rpos1 + (((qpos & SPARSE_MASK) * (rpos2 - rpos1)) >> SPARSE_STEP_PWR)

    def find_cigar_offset(self, qpos, approx, ref_to_query):
        min_ix, max_ix, _ = approx
        qsel = (lambda p: p[1]) if ref_to_query else (lambda p: p[0])
        rsel = (lambda p: p[0]) if ref_to_query else (lambda p: p[1])
        if min_ix == max_ix: ix = min_ix
        else:
            lo, hi = min_ix, max_ix + 1                      # bisect::right_by_at, algo/bisect.rs:72-83
            while lo < hi:
                mid = (lo + hi) // 2
                if qsel(self.positions[mid]) > qpos: hi = mid
                else: lo = mid + 1
            ix = lo - 1
        p = self.positions[ix]
        return ix, qsel(p), rsel(p)


def align_simple(seq1, seq2, cig):                           # wfa.rs:49-84
    n, m = len(seq1), len(seq2)
    diff = n - m
    if diff < 0: cig.push_unchecked("I", -diff); score = -GAP_OPEN + diff * GAP_EXTEND; i, j = 0, -diff
    elif diff > 0: cig.push_unchecked("D", diff); score = -GAP_OPEN - diff * GAP_EXTEND; i, j = diff, 0
    else: score, i, j = 0, 0, 0
    curr_match, curr_len = seq1[i] == seq2[j], 1
    for a, b in zip(seq1[i + 1:], seq2[j + 1:]):
        if (a == b) != curr_match:
            cig.push_unchecked("=" if curr_match else "X", curr_len)
            score -= 0 if curr_match else MISMATCH * curr_len
            curr_match, curr_len = not curr_match, 1
        else: curr_len += 1
    cig.push_unchecked("=" if curr_match else "X", curr_len)
    score -= 0 if curr_match else MISMATCH * curr_len
    return score


def wfa_align(left_clipping, mode, seq1, seq2, cig):         # Aligner::align, wfa.rs:254-299; mode 0 global, 1 LEFT free, 2 RIGHT free
    pen, ops = O.dp_align(bytes(seq1), bytes(seq2), max(1, MISMATCH // 2) if mode else 0, mode)
    if pen == DROPPED:
        return align_simple(seq1, seq2, cig)
    no_matches_yet = True
    for op, ln in Cig.parse(ops).t:
        for _ in range(ln):
            if left_clipping and no_matches_yet and op == "=":
                no_matches_yet = False
                soft = cig.qlen
                cig.clear()
                if soft > 0: cig.push_unchecked("I", soft)
            cig.push_checked(op, 1)
    if left_clipping and no_matches_yet:
        soft = cig.qlen
        cig.clear()
        if soft > 0: cig.push_unchecked("I", soft)
    return -pen


def smart_align(seq1, i1, i2, seq2, j1, j2, max_gap, cig):  # wfa.rs:301-347; max_gap None = the `()` threshold
    jump1, jump2 = i2 - i1, j2 - j1
    if jump1 > 0 and jump2 > 0:
        s1, s2 = seq1[i1:i2], seq2[j1:j2]
        if max_gap is not None and (max_gap < jump1 or max_gap < jump2): return align_simple(s1, s2, cig)
        if jump1 == jump2 and jump1 <= SAFE_MISMATCH:
            nd = 0
            for a, b in zip(s1, s2):
                cig.push_checked("=" if a == b else "X", 1); nd -= a != b
            return nd * MISMATCH
        return wfa_align(False, 0, s1, s2, cig)
    if jump1 > 0: cig.push_unchecked("D", jump1); return -GAP_OPEN - jump1 * GAP_EXTEND
    if jump2 > 0: cig.push_unchecked("I", jump2); return -GAP_OPEN - jump2 * GAP_EXTEND
    return 0


def align_ends(left, seq1, i1, i2, seq2, j1, j2, cig):      # wfa.rs:349-365
    assert j1 != j2
    if i1 == i2:
        cig.push_unchecked("I", j2 - j1); return
    wfa_align(left, 1 if left else 2, seq1[i1:i2], seq2[j1:j2], cig)
    if not left:
        soft = 0
        while cig.t and cig.t[-1][0] != "=":
            op, ln = cig.t.pop()
            q, r = consumes(op)
            cig.qlen -= ln if q else 0; cig.rlen -= ln if r else 0
            soft += ln if q else 0
        if soft > 0: cig.push_unchecked("I", soft)


def optimize(self, ref_seq, query_seq, max_gap, anchor_size):               # cigar.rs:1167-1237
    i = qpos1 = rpos1 = flag = qpos2 = rpos2 = 0
    new = None
    for j, (op, ln) in enumerate([tuple(x) for x in self.t]):
        cq, cr = consumes(op)
        if cq and cr and ln >= anchor_size:
            qshift, rshift = qpos2 - qpos1, rpos2 - rpos1
            if flag == 3 and not max_gap < qshift and not max_gap < rshift:
                if new is None:
                    new = Cig(); new.t = [list(x) for x in self.t[:i]]; new.qlen, new.rlen = qpos1, rpos1
                smart_align(ref_seq, rpos1, rpos2, query_seq, qpos1, qpos2, None, new)
                i = j
            qpos2 += ln; rpos2 += ln; qpos1, rpos1, flag = qpos2, rpos2, 0
            if new is not None:
                new.t.extend([list(x) for x in self.t[i:j]])
                new.push_checked(op, ln)
                new.qlen, new.rlen = qpos2, rpos2
            i = j + 1
        else:
            qpos2 += ln if cq else 0; rpos2 += ln if cr else 0
            flag |= (0 if cq else 1) | ((0 if cr else 1) << 1)
    qshift, rshift = qpos2 - qpos1, rpos2 - rpos1
    if flag == 3 and not max_gap < qshift and not max_gap < rshift:
        if new is None:
            new = Cig(); new.t = [list(x) for x in self.t[:i]]; new.qlen, new.rlen = qpos1, rpos1
        smart_align(ref_seq, rpos1, rpos2, query_seq, qpos1, qpos2, None, new)
        i = len(self.t)
    if new is not None:
        new.t.extend([list(x) for x in self.t[i:]])
        self.t = new.t                                       # the lengths stay as they were


def double_move(op1, op2, st):                               # cigar.rs:1422-1466; st = [pos1, rem1, pos2, rem2]
    cls = lambda op: {(True, True): "B", (True, False): "Q", (False, True): "R"}[consumes(op)]
    read_moves, read_shifts, hap_moves, hap_shifts = {
        ("B", "B"): (1, 1, 1, 1), ("Q", "B"): (1, 1, 0, 0), ("R", "B"): (0, 1, 1, 1),
        ("B", "Q"): (1, 1, 0, 1), ("Q", "Q"): (1, 1, 0, 0), ("R", "Q"): (0, 1, 0, 1),
        ("B", "R"): (0, 0, 1, 1), ("Q", "R"): (1, 1, 1, 1), ("R", "R"): (0, 0, 1, 1)}[(cls(op1), cls(op2))]
    shift = st[1] if read_shifts and (not hap_shifts or st[1] <= st[3]) else st[3]
    st[0] += shift if read_moves else 0; st[1] -= shift if read_shifts else 0
    st[2] += shift if hap_moves else 0; st[3] -= shift if hap_shifts else 0
    return shift


def transfer_read_alignment(hap_cig, index, ref_to_query, start_j, read_cig, seq_i, seq_k):
    """Cigar::transfer_alignment::<false> with ANCHOR_SIZE 5, no maximum gap (cigar.rs:1248-1384). Returns (start_k, Cig)."""
    anchor_size, FULL_MATCH_PADDING, CLIP_PADDING, ANCHOR_MARGIN = 5, 3, 3, 5
    off_ix, off_qpos, off_rpos = index.find_cigar_offset(start_j, index.find_approx_position(start_j, ref_to_query), ref_to_query)
    dir_op = invert_op if ref_to_query else (lambda o: o)
    jk = iter(hap_cig.t[off_ix:])
    op2, len2 = next(jk); op2 = dir_op(op2)
    init_shift = start_j - off_qpos
    rem2 = len2 - init_shift
    start_k = off_rpos + (init_shift if consumes(op2)[1] else 0)
    len_k = len(seq_k)
    if op2 == "=" and init_shift >= FULL_MATCH_PADDING and rem2 >= read_cig.rlen + FULL_MATCH_PADDING:
        return start_k, Cig([tuple(x) for x in read_cig.t])
    ij = iter(read_cig.t)
    op1, len1 = next(ij); rem1 = len1
    last1 = 0; last2 = start_k
    st = [0, rem1, start_k, rem2]                            # pos1, rem1, pos2, rem2
    new = Cig()
    while True:
        pos1, rem1, pos2, rem2 = st
        e1, e2 = op1 == "=", op2 == "="
        add = None
        if e1 and e2:
            if min(rem1, rem2) >= anchor_size: add = "="
        elif e1 and not e2:
            if rem1 >= anchor_size and len1 - rem1 >= ANCHOR_MARGIN: add = op2
        elif not e1 and e2:
            if rem2 >= anchor_size and len2 - rem2 >= ANCHOR_MARGIN: add = op1
        if add is not None:
            if last1 == 0 and pos1 > 0:
                align_ends(True, seq_k, max(0, last2 - (pos1 + CLIP_PADDING)), pos2, seq_i, last1, pos1, new)
                start_k = start_k + pos2 - last2 - new.rlen
            else:
                smart_align(seq_k, last2, pos2, seq_i, last1, pos1, None, new)
        shift = double_move(op1, op2, st)
        if add is not None:
            new.push_checked(add, shift); last1, last2 = st[0], st[2]
        if st[1] == 0:
            nxt = next(ij, None)
            if nxt is None: break
            op1, len1 = nxt; st[1] = len1
        if st[3] == 0:
            nxt = next(jk, None)
            if nxt is None: break
            op2, len2 = nxt; op2 = dir_op(op2); st[3] = len2
    len_i = len(seq_i)
    if last1 != len_i:
        align_ends(False, seq_k, last2, min(len_k, last2 + len_i - last1 + CLIP_PADDING), seq_i, last1, len_i, new)
    assert new.qlen == len_i, (new.qlen, len_i)
    optimize(new, seq_k, seq_i, 20, 5)
    if new.t[0][0] == "I": new.t[0][0] = "S"               # boundary_ins_to_soft, cigar.rs:554-561
    if new.t[-1][0] == "I": new.t[-1][0] = "S"
    return start_k, new
