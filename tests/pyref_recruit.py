"""Second, independent transliteration of minimizer read recruitment (TEST INFRASTRUCTURE), line by line from the Rust
(seq/kmers.rs:93-103, 243-331; seq/recruit.rs:236-367, 688-738, 848-996; math/frac.rs:50-93), written separately from
oracle/lcty_oracle_recruit.c; tests/test_oracle_recruit.py holds the two against each other."""
import math

M64 = (1 << 64) - 1
UNDEF = M64
SUBSUM_BONUS, SUBSUM_PENALTY, READ_LENGTH_THRESH, WORTH = 3, 1, 500, 3


def fast_hash(x):                                            # kmers.rs:93-103
    x = ~x & M64
    x ^= x >> 23
    x = (x * 0x2127599bf4325c37) & M64
    x ^= x >> 47
    return x


def canon_minimizers(seq, k, w):
    """[(pos, hash, forward)] — kmers.rs:265-331"""
    mask = (1 << (2 * k)) - 1
    rv_shift = 2 * k - 2
    fw_kmer = rv_kmer = 0
    k_1, w_1 = k - 1, w - 1
    hashes, forward = [UNDEF] * 64, [True] * 64
    last_pos, best_pos, best_hash = -1, 0, UNDEF
    first_kmer, first_window = k_1, k_1 + w_1
    out = []
    for i, nt in enumerate(seq):
        enc = {65: (0, 3), 67: (1, 2), 71: (2, 1), 84: (3, 0)}.get(nt)
        if enc is None:
            first_kmer = i + k
            enc = (0, 0)
        fw_kmer = ((fw_kmer << 2) | enc[0]) & mask
        rv_kmer = (rv_kmer >> 2) | (enc[1] << rv_shift)
        kmer, fw = (rv_kmer, False) if rv_kmer < fw_kmer else (fw_kmer, True)
        h = UNDEF if i < first_kmer else fast_hash(kmer)
        hashes[i & 63], forward[i & 63] = h, fw
        if h < best_hash: best_hash, best_pos = h, i
        if i < first_window: continue
        start = i - w_1
        if best_pos < start:
            best_pos, best_hash = start, hashes[start & 63]
            for j in range(start + 1, i + 1):
                if hashes[j & 63] < best_hash: best_pos, best_hash = j, hashes[j & 63]
            if best_hash == UNDEF:
                first_window = first_window + w_1
                continue
        if best_pos > last_pos:
            last_pos = best_pos
            out.append((best_pos - k_1, best_hash, forward[best_pos & 63]))
    return out


def approximate_u16(x):                                      # frac.rs:50-76
    a2, a1, b2, b1 = 1, int(math.floor(x)), 0, 1
    xk = x
    for _ in range(20):
        numer = xk - math.floor(xk)
        if numer <= 2.220446049250313e-16: break
        xk = 1.0 / numer
        fl = math.floor(xk)
        if not 0 <= fl <= 65535: break
        fl = int(fl)
        if fl * a1 > 65535 or fl * a1 + a2 > 65535 or fl * b1 > 65535 or fl * b1 + b2 > 65535: break
        a2, a1, b2, b1 = a1, fl * a1 + a2, b1, fl * b1 + b2
        if abs(a1 / b1 - x) <= 2.220446049250313e-16: break
    return a1, b1


class Targets:
    def __init__(self, k=15, w=10, match_frac=0.5, match_length=2000, thresh_kmer_count=50):
        self.k, self.w, self.match_frac, self.match_length, self.thresh = k, w, match_frac, match_length, thresh_kmer_count
        self.stretch_minims = (2 * match_length + (w + 1) - 1) // (w + 1)
        self.stretch_score = int(math.ceil(max(self.stretch_minims * ((SUBSUM_BONUS + SUBSUM_PENALTY) * match_frac - SUBSUM_PENALTY), SUBSUM_BONUS)))
        self.mf = approximate_u16(match_frac)
        self.minim_to_loci = {}                              # minimizer -> [[locus, direction, rare]]
        self.locus_minimizers = []

    def add(self, seqs, counts, base_k):                     # TargetBuilder::add, recruit.rs:688-738
        locus = len(self.locus_minimizers)
        shift = (base_k - self.k) // 2 if self.k <= base_k else self.k - base_k
        lm = {}
        for seq, cnt in zip(seqs, counts):
            for pos, minim, fw in canon_minimizers(seq, self.k, self.w):
                if self.k <= base_k: rare = cnt[min(max(pos - shift, 0), len(cnt) - 1)] < self.thresh
                else: rare = cnt[pos] < self.thresh and cnt[pos + shift] < self.thresh
                v = self.minim_to_loci.setdefault(minim, [])
                if v and v[-1][0] == locus: v[-1][1] |= 1 + fw; v[-1][2] &= rare
                else: v.append([locus, 1 + fw, rare])
                e = lm.setdefault(minim, [0, True])
                e[0] |= 1 + fw; e[1] &= rare
        self.locus_minimizers.append(lm)
        return locus

    @staticmethod
    def _inc(arr, forward, direction, rare):                 # BaseMatchCount::inc, recruit.rs:250-256
        i = int(rare) << 1
        arr[i] += (direction & (1 + (not forward))) != 0
        arr[i | 1] += (direction & (1 + forward)) != 0

    def _ge(self, num, den):                                 # Fraction<u16> >= match_frac_short, frac.rs:87-93
        return (num & 0xFFFF) * self.mf[1] >= self.mf[0] * (den & 0xFFFF)

    def recruit(self, seq1, seq2=None):
        m1 = canon_minimizers(seq1, self.k, self.w)
        total1 = len(m1)
        matches = {}
        for _, minim, fw in m1:
            for locus, direction, rare in self.minim_to_loci.get(minim, ()):
                self._inc(matches.setdefault(locus, ([0] * 4, [0] * 4))[0], fw, direction, rare)
        fn = lambda a: WORTH * a[3] + a[1]
        bn = lambda a: WORTH * a[2] + a[0]
        fd = lambda a, t: WORTH * (t - a[1]) + a[1]
        bd = lambda a, t: WORTH * (t - a[0]) + a[0]
        ans = []
        if seq2 is not None:                                 # recruit_read_pair, 883-929
            if not matches: return []
            m2 = canon_minimizers(seq2, self.k, self.w)
            total2 = len(m2)
            for _, minim, fw in m2:
                for locus, direction, rare in self.minim_to_loci.get(minim, ()):
                    if locus in matches: self._inc(matches[locus][1], fw, direction, rare)
            for locus, (a, b) in matches.items():
                if a[2] or a[3] or b[2] or b[3]:
                    if (fn(a) + bn(b)) & 0xFFFF >= (bn(a) + fn(b)) & 0xFFFF: f1, f2 = (fn(a), fd(a, total1)), (bn(b), bd(b, total2))
                    else: f1, f2 = (bn(a), bd(a, total1)), (fn(b), fd(b, total2))
                    if self._ge(*f1) and self._ge(*f2): ans.append(locus)
        elif len(seq1) <= READ_LENGTH_THRESH:                # recruit_short_read, 848-879
            for locus, (a, _) in matches.items():
                if a[2] or a[3]:
                    f = (fn(a), fd(a, total1)) if fn(a) >= bn(a) else (bn(a), bd(a, total1))
                    if self._ge(*f): ans.append(locus)
        else:                                                # recruit_long_read, 964-996
            for locus, (a, _) in matches.items():
                num, den = (a[3], total1 - a[1]) if a[3] >= a[2] else (a[2], total1 - a[0])
                thr = max(1, int(math.ceil(min(self.stretch_minims, den) * self.match_frac)))
                if num >= thr and (den < self.stretch_minims or self._stretch(locus, m1)): ans.append(locus)
        return sorted(ans)

    def _stretch(self, locus, minims):                       # has_matching_stretch, 938-961
        lm = self.locus_minimizers[locus]
        s_fw = s_bw = 0
        for _, minim, fw in minims:
            e = lm.get(minim)
            if e is not None:
                x = SUBSUM_PENALTY + int(e[1]) * SUBSUM_BONUS
                s_fw += ((e[0] & (1 + fw)) != 0) * x
                s_bw += ((e[0] & (1 + (not fw))) != 0) * x
            s_fw, s_bw = max(s_fw - SUBSUM_PENALTY, 0), max(s_bw - SUBSUM_PENALTY, 0)
            if s_fw >= self.stretch_score or s_bw >= self.stretch_score: return True
        return False
