"""lcty_write_bam (write_bam, src/model/bam.rs:356-413): the per-genotype BAM of read placements with posteriors from the assignment
counts, read back with an independent parser: header, sort order, flags and mate fields, tags, the fold of locations by alignment pair,
MAPQ / pr from the counts, the BAI index."""
import gzip
import math
import struct

import numpy as np
import pytest

from locityper_amd import _lib, api, cdefs, io as lio, synth
from tests import oracle_ffi as O

pytestmark = pytest.mark.gpu
NT16 = "=ACMGRSVTWYHKDBN"


def parse_bam(path):
    b = gzip.open(path, "rb").read()
    assert b[:4] == b"BAM\x01"
    l_text, = struct.unpack_from("<I", b, 4)
    text = b[8:8 + l_text].decode()
    i = 8 + l_text
    n_ref, = struct.unpack_from("<I", b, i); i += 4
    refs = []
    for _ in range(n_ref):
        ln, = struct.unpack_from("<I", b, i); i += 4
        nm = b[i:i + ln - 1].decode(); i += ln
        rl, = struct.unpack_from("<I", b, i); i += 4
        refs.append((nm, rl))
    recs = []
    while i < len(b):
        bs, = struct.unpack_from("<I", b, i); i += 4
        tid, pos, l_name, mapq, bin_, n_cig, flag, l_seq, mtid, mpos, tlen = struct.unpack_from("<iiBBHHHIiii", b, i)
        j = i + 32
        name = b[j:j + l_name - 1].decode(); j += l_name
        cigar = list(struct.unpack_from(f"<{n_cig}I", b, j)); j += 4 * n_cig
        packed = b[j:j + (l_seq + 1) // 2]; j += (l_seq + 1) // 2
        seq = "".join(NT16[(packed[k >> 1] >> (4 if k % 2 == 0 else 0)) & 15] for k in range(l_seq))
        qual = b[j:j + l_seq]; j += l_seq
        tags = {}
        while j < i + bs:
            tag, ty = b[j:j + 2].decode(), chr(b[j + 2]); j += 3
            fmt = {"I": "<I", "f": "<f", "S": "<H", "A": "<c"}[ty]
            v, = struct.unpack_from(fmt, b, j); j += struct.calcsize(fmt)
            tags[tag] = v.decode() if ty == "A" else v
        recs.append(dict(tid=tid, pos=pos, mapq=mapq, bin=bin_, flag=flag, mtid=mtid, mpos=mpos, tlen=tlen, name=name, cigar=cigar,
                         seq=seq, qual=qual, tags=tags))
        i += bs
    return text, refs, recs


def revcomp(s):
    return s[::-1].translate(str.maketrans("ACGTN", "TGCAN"))


@pytest.mark.parametrize("paired", [True, False])
def test_bam_of_one_genotype(gpu_ctx, tmp_path, paired):
    n_alleles, n_pairs, attempts = 6, 1500, 8
    tech = cdefs.TECH_ILLUMINA if paired else cdefs.TECH_NANOPORE
    L = synth.SynthLocus(n_alleles, n_pairs, base_len=12_000, technology=tech, read_len=150 if paired else 2000)
    p = api.resolve_params(api.default_params(), L.bg)
    loc = api.Locus(gpu_ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    ch = L.reads(0, n_pairs)
    aa = api.AllAlignments.load(loc, ch)
    gt = np.array(L.true_genotype, dtype=np.uint16)
    seeds = api.chain_seeds(9, attempts)
    read_off, counts = api.assignment_counts(aa, gt, api.default_solver(cdefs.SOLVER_ANNEAL), attempts, seeds)
    names = [f"read{r}" for r in range(n_pairs)]
    allele_names = [f"hap{a}" for a in range(n_alleles)]
    quals = [bytes([(7 * m + k) % 40 for k in range(int(ch.mate_len[m]))]) for m in range(2 * n_pairs)]
    path = tmp_path / "00.bam"
    n_written = lio.write_bam(path, aa, ch, names, allele_names, gt, attempts, read_off, counts, quals=quals)
    text, refs, recs = parse_bam(path)
    uniq = list(dict.fromkeys(int(a) for a in gt))
    assert refs == [(allele_names[a], int(loc.allele_len[a])) for a in uniq] and n_written == len(recs)
    assert text == "".join(f"@SQ\tSN:{allele_names[a]}\tLN:{int(loc.allele_len[a])}\n" for a in uniq)
    # coordinate-sorted, unmapped templates at the end
    keys = [(r["tid"] & 0xFFFFFFFF, r["pos"]) for r in recs]
    assert keys == sorted(keys)
    # records of one position: those of the used reads before those of the unused ones (bam.rs:372-398 pushes all used reads first, the sort is stable)
    for x, y in zip(recs, recs[1:]):
        if (x["tid"], x["pos"]) == (y["tid"], y["pos"]):
            assert not (x["tags"]["us"] == "F" and y["tags"]["us"] == "T")
    status, weight, unm, uk = aa.status()
    by_read = {}
    for r in recs:
        by_read.setdefault(int(r["name"][4:]), []).append(r)
    good = np.flatnonzero(status == cdefs.READ_GOOD)
    assert set(by_read) <= set(np.flatnonzero((status == cdefs.READ_GOOD) | (status == cdefs.READ_FEW_KMERS)).tolist())
    assert set(good.tolist()) <= set(by_read)
    lp = [math.log(x) if False else x for x in L.bg.op_lnprobs]
    for gi, r in enumerate(good[:400]):
        rr = by_read[int(r)]
        first = [x for x in rr if not paired or x["flag"] & 0x40]
        assert all(x["tags"]["us"] == "T" for x in rr)
        # the folds of a read carry all its attempts; exactly one is primary; MAPQ from the probability (count_to_prob)
        c = counts[int(read_off[gi]):int(read_off[gi + 1])]
        assert int(c.sum()) == attempts and abs(sum(x["tags"]["pr"] for x in first) - 1.0) < 1e-6
        assert sum(1 for x in first if not x["flag"] & 0x100) == 1
        prim = [x for x in first if not x["flag"] & 0x100][0]
        assert prim["tags"]["pr"] == max(x["tags"]["pr"] for x in first)
        for x in rr:
            pr = x["tags"]["pr"]
            want = 60 if pr == 1.0 else int(min(round(-10.0 * math.log10(1.0 - pr)), 60.0))
            assert x["mapq"] == want and x["tags"]["uk"] == int(uk[2 * r + (1 if paired and x["flag"] & 0x80 else 0)])
            if not x["flag"] & 0x4:
                ops = [(w & 15, w >> 4) for w in x["cigar"]]
                ref = sum(n for o, n in ops if o in (7, 8, 2))
                clen = refs[x["tid"]][1]
                left = ops[0][1] if ops[0][0] == 4 else 0
                right = ops[-1][1] if ops[-1][0] == 4 and len(ops) > 1 else 0
                clip = min(left, x["pos"]) + min(right, max(clen - x["pos"] - ref, 0))
                assert x["tags"]["NM"] == sum(n for o, n in ops if o in (8, 1, 2)) + clip
                assert x["bin"] == _reg2bin(x["pos"], x["pos"] + max(ref, 1))
                assert len(x["seq"]) == sum(n for o, n in ops if o in (7, 8, 1, 4))
            if paired:
                mate = [y for y in rr if y is not x and y["tags"]["pr"] == pr and (y["flag"] & 0xC0) != (x["flag"] & 0xC0)
                        and y["mpos"] == x["pos"] and y["mtid"] == x["tid"]]
                assert mate and x["flag"] & 0x1 and x["tlen"] == -mate[0]["tlen"]
                assert bool(x["flag"] & 0x20) == bool(mate[0]["flag"] & 0x10) or (x["flag"] & 0x4) or (mate[0]["flag"] & 0x4)
        # sequence and qualities: as stored for the strand of the primary record, reverse-complemented on the other strand
        for x in first[:1]:
            m = 2 * int(r)
            stored = "".join("N" if (ch.nmask[(int(ch.mate_off[m]) + k) >> 5] >> ((int(ch.mate_off[m]) + k) & 31)) & 1 else
                             "ACGT"[(int(ch.bases2[(int(ch.mate_off[m]) + k) >> 4]) >> (2 * ((int(ch.mate_off[m]) + k) & 15))) & 3]
                             for k in range(int(ch.mate_len[m])))
            stored_rev = bool(int(ch.recs[int(ch.aln_off[r])]["flags"]) & cdefs.FLAG_REVERSE)
            rev = bool(x["flag"] & 0x10) if not x["flag"] & 0x4 else False
            assert x["seq"] == (stored if rev == stored_rev else revcomp(stored))
            assert x["qual"] == (quals[m] if rev == stored_rev else quals[m][::-1])
    # reads with few unique k-mers: us:F, no pr
    for r in np.flatnonzero(status == cdefs.READ_FEW_KMERS)[:50]:
        for x in by_read.get(int(r), []):
            assert x["tags"]["us"] == "F" and "pr" not in x["tags"] and x["mapq"] == 0
    # the index: references, the pseudo-bin's counts, the number of records without coordinates
    bai = open(str(path) + ".bai", "rb").read()
    assert bai[:4] == b"BAI\x01" and struct.unpack_from("<I", bai, 4)[0] == len(refs)
    assert struct.unpack_from("<Q", bai, len(bai) - 8)[0] == sum(1 for x in recs if x["tid"] < 0)
    # the container is what the library's own reader takes (BGZF with the end-of-file block)
    raw = lio.read_file(path)
    assert raw[:4] == b"BAM\x01" and open(path, "rb").read()[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    # counts of another genotype are refused, so is a batch whose table has been replaced by alignment recovery
    other = np.array([0, 0], dtype=np.uint16) if tuple(gt) != (0, 0) else np.array([1, 1], dtype=np.uint16)
    ro2, c2 = api.assignment_counts(aa, other, api.default_solver(cdefs.SOLVER_GREEDY), 1, api.chain_seeds(1, 1))
    if len(c2) != len(counts):
        with pytest.raises(_lib.LocityperError):
            lio.write_bam(tmp_path / "x.bam", aa, ch, names, allele_names, gt, 1, ro2, c2)


def test_bam_after_alignment_recovery(gpu_ctx, tmp_path):
    """The mapper reported the primaries only; the other alleles came in through alignment recovery. The batch's record table comes
    back from the device (lcty_reads_get_records) and lcty_write_bam places the reads on the call with it: the transferred alignments
    are records like any other."""
    n_alleles, n_pairs, attempts = 6, 1200, 6
    L = synth.SynthLocus(n_alleles, n_pairs, base_len=12_000)
    p = api.resolve_params(api.default_params(), L.bg)
    loc = api.Locus(gpu_ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    loc.set_hap_alns(L.hap_alns(), transfer_fails=100, max_div=0.1)
    ch = L.reads(0, n_pairs, primaries_only=True)
    aa = api.AllAlignments.load(loc, ch)
    n_new = aa.recover()
    aa.score()
    assert n_new > 3 * n_pairs
    aln_off, recs, cig_off, cigar = aa.records()
    assert len(recs) == len(ch.recs) + n_new and np.all(np.diff(aln_off.astype(np.int64)) >= np.diff(ch.aln_off.astype(np.int64)))
    merged = cdefs.ReadsChunk(ch.mate_len, ch.mate_off, ch.bases2, ch.nmask, aln_off, recs, cig_off, cigar)
    gt = np.array(L.true_genotype, dtype=np.uint16)
    read_off, counts = api.assignment_counts(aa, gt, api.default_solver(cdefs.SOLVER_ANNEAL), attempts, api.chain_seeds(9, attempts))
    names = [f"read{r}" for r in range(n_pairs)]
    allele_names = [f"hap{a}" for a in range(n_alleles)]
    path = tmp_path / "00.bam"
    with pytest.raises(_lib.LocityperError):
        lio.write_bam(path, aa, ch, names, allele_names, gt, attempts, read_off, counts)      # the mapper's table no longer matches
    n_written = lio.write_bam(path, aa, merged, names, allele_names, gt, attempts, read_off, counts)
    text, refs, bam = parse_bam(path)
    assert n_written == len(bam) and len(refs) == len(set(L.true_genotype))
    status = aa.status()[0]
    good = np.flatnonzero(status == cdefs.READ_GOOD)
    by_read = {}
    for r in bam:
        by_read.setdefault(int(r["name"][4:]), []).append(r)
    assert set(good.tolist()) <= set(by_read) and len(good) > n_pairs // 2
    on_other = 0
    for gi, r in enumerate(good[:300]):
        rr = by_read[int(r)]
        first = [x for x in rr if x["flag"] & 0x40]
        assert abs(sum(x["tags"]["pr"] for x in first) - 1.0) < 1e-6
        for x in rr:
            if x["flag"] & 0x4: continue
            ops = [(w & 15, w >> 4) for w in x["cigar"]]
            assert len(x["seq"]) == sum(n for o, n in ops if o in (7, 8, 1, 4)) == int(ch.mate_len[2 * r + (1 if x["flag"] & 0x80 else 0)])
            assert 0 <= x["pos"] and x["pos"] + sum(n for o, n in ops if o in (7, 8, 2)) <= refs[x["tid"]][1]
            # a placement on an allele the mapper had no record on can only be a transferred alignment
            src = int(ch.recs[int(ch.aln_off[r])]["contig"])
            on_other += allele_names[src] != refs[x["tid"]][0]
    assert on_other > 50


def _reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0
