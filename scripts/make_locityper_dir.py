#!/usr/bin/env python3
"""Writes one synthetic locus in the directory layout `locityper genotype` works on, for examples/genotype_dir.cpp:

    ROOT/DB/loci/<locus>/haplotypes.fa.gz     alleles a0, a1, ..
    ROOT/DB/loci/<locus>/kmers.bin.lz4        two KmerCounts blocks (seq/counts.rs:108-150): off-target counts, then regular counts
    ROOT/PREPROC/distr.gz                     BgDistr as JSON (bg/mod.rs:147-177), pretty-printed as the reference writes it
    ROOT/OUT/loci/<locus>/aln.bam             name-grouped records, mates mapped as independent reads (genotype.rs:975-977)
    ROOT/truth.json                           the genotype the reads were drawn from

usage: make_locityper_dir.py ROOT [--locus L1 --alleles 8 --pairs 10000 --base-len 50000]"""
import argparse
import gzip
import json
import os
import struct
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locityper_amd import cdefs, synth  # noqa: E402

NT16 = "=ACMGRSVTWYHKDBN"


def varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F; v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def kmer_counts_block(k, contigs):
    out = bytearray([k, 2]) + varint(len(contigs))
    for c in contigs:
        out += varint(len(c))
        out += b"".join(varint(int(x)) for x in c)
    return bytes(out)


def lz4_frame_stored(data, block=1 << 22):
    """An LZ4 frame of uncompressed blocks (valid LZ4: the high bit of a block size marks stored data)."""
    out = struct.pack("<I", 0x184D2204) + bytes([0x60, 0x70, 0x73])      # version 01, independent blocks; 4 MB blocks; HC unchecked by readers here
    for i in range(0, len(data), block):
        chunk = data[i:i + block]
        out += struct.pack("<I", len(chunk) | 0x80000000) + chunk
    return out + struct.pack("<I", 0)


def bgzf(data):
    out = bytearray()
    for i in list(range(0, len(data), 0xff00)) + [None]:
        chunk = b"" if i is None else data[i:i + 0xff00]
        comp = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = comp.compress(chunk) + comp.flush()
        out += struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(body) + 25) + body
        out += struct.pack("<II", zlib.crc32(chunk), len(chunk))
    return bytes(out)


def unpack(chunk, mate):
    off, ln = int(chunk.mate_off[mate]), int(chunk.mate_len[mate])
    idx = off + np.arange(ln)
    codes = (chunk.bases2[idx >> 4] >> (2 * (idx & 15))) & 3
    isn = (chunk.nmask[idx >> 5] >> (idx & 31)) & 1
    return "".join("N" if n else "ACGT"[c] for c, n in zip(codes, isn))


def bam_bytes(names_lens, chunk, first_read=0):
    text = b"@HD\tVN:1.6\tSO:unsorted\tGO:query\n"
    out = bytearray(b"BAM\x01" + struct.pack("<I", len(text)) + text + struct.pack("<I", len(names_lens)))
    for nm, ln in names_lens:
        out += struct.pack("<I", len(nm) + 1) + nm.encode() + b"\0" + struct.pack("<I", ln)
    for r in range(chunk.n_pairs):
        lo, hi = int(chunk.aln_off[r]), int(chunk.aln_off[r + 1])
        cbase, end = int(chunk.cigar_off[r]), 0
        qname = f"read{first_read + r}".encode()
        for t in range(lo, hi):
            rec = chunk.recs[t]
            flags = int(rec["flags"]) & ~cdefs.FLAG_MATE2                  # the mapper sees single-end reads
            primary = flags & (cdefs.FLAG_SECONDARY | cdefs.FLAG_SUPPL) == 0
            if primary and t > lo:
                end = 1
            c0 = cbase + int(rec["cigar_rel"])
            words = chunk.cigar[c0:c0 + int(rec["n_cigar"])]
            seq = unpack(chunk, 2 * r + end) if primary else ""
            packed = bytearray((len(seq) + 1) // 2)
            for k, ch in enumerate(seq):
                packed[k >> 1] |= NT16.index(ch) << (4 if k % 2 == 0 else 0)
            unm = bool(flags & cdefs.FLAG_UNMAPPED)
            body = struct.pack("<iiBBHHHIiii", -1 if unm else int(rec["contig"]), -1 if unm else int(rec["pos"]), len(qname) + 1, 0 if unm else 30, 4680,
                               len(words), flags, len(seq), -1, -1, 0)
            body += qname + b"\0" + words.astype("<u4").tobytes() + bytes(packed) + bytes([30] * len(seq))
            out += struct.pack("<I", len(body)) + body
    return bytes(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--locus", default="L1")
    ap.add_argument("--alleles", type=int, default=8)
    ap.add_argument("--pairs", type=int, default=10_000)
    ap.add_argument("--base-len", type=int, default=50_000)
    ap.add_argument("--paf", action="store_true", help="the layout of a run on basis haplotypes: aln.bam holds the primary alignments only and "
                    "DB/loci/<locus>/haplotypes.paf.gz the pairwise haplotype alignments that carry them to the other alleles")
    a = ap.parse_args()
    L = synth.SynthLocus(a.alleles, a.pairs, base_len=a.base_len)
    db, outd = os.path.join(a.root, "DB", "loci", a.locus), os.path.join(a.root, "OUT", "loci", a.locus)
    os.makedirs(db, exist_ok=True); os.makedirs(os.path.join(outd, "alns"), exist_ok=True); os.makedirs(os.path.join(a.root, "PREPROC"), exist_ok=True)
    names = [f"a{i}" for i in range(a.alleles)]
    lens = [int(L.seq_off[i + 1] - L.seq_off[i]) for i in range(a.alleles)]
    with gzip.open(os.path.join(db, "haplotypes.fa.gz"), "wt") as f:
        for i, nm in enumerate(names):
            s = L.allele(i).decode()
            f.write(f">{nm} synthetic allele {i}\n")
            f.writelines(s[j:j + 80] + "\n" for j in range(0, len(s), 80))
    off = [L.counts[int(L.cnt_off[i]):int(L.cnt_off[i + 1])] for i in range(a.alleles)]
    regular = [np.ones(len(c), dtype=np.uint16) for c in off]
    open(os.path.join(db, "kmers.bin.lz4"), "wb").write(lz4_frame_stored(kmer_counts_block(L.k, off) + kmer_counts_block(L.k, regular)))
    bg = L.bg
    tech = {cdefs.TECH_ILLUMINA: "illumina", cdefs.TECH_HIFI: "hifi", cdefs.TECH_PACBIO: "pacbio", cdefs.TECH_NANOPORE: "nanopore"}[bg.technology]
    distr = {"seq_info": {"read_len": 150.0, "technology": tech, "total_reads": 2 * a.pairs, "file_size": None},
             "insert_distr": {"n": bg.ins_n, "p": bg.ins_p} if bg.is_paired else {},
             "error_profile": {"matches": bg.op_lnprobs[0], "mismatches": bg.op_lnprobs[1], "insertions": bg.op_lnprobs[2],
                               "deletions": bg.op_lnprobs[3], "clipping": bg.op_lnprobs[4], "alpha": bg.edit_alpha, "beta": bg.edit_beta},
             "bg_depth": {"ploidy": 2, "window": bg.window, "neighb": bg.neighb, "n": list(bg.depth_n), "p": list(bg.depth_p)}}
    with gzip.open(os.path.join(a.root, "PREPROC", "distr.gz"), "wt") as f:
        json.dump(distr, f, indent=4)
    ch = L.reads(0, a.pairs, primaries_only=a.paf)
    if a.paf:
        with gzip.open(os.path.join(db, "haplotypes.paf.gz"), "wt") as f:
            for q, t, words, nm, al in L.hap_alns():
                cg = "".join(f"{int(w) >> 4}{'MIDNSHP=X'[int(w) & 15]}" for w in words)
                f.write(f"{names[q]}\t{lens[q]}\t0\t{lens[q]}\t+\t{names[t]}\t{lens[t]}\t0\t{lens[t]}\t{nm}\t{al}\t60\ttp:A:P\tcg:Z:{cg}\n")
    open(os.path.join(outd, "aln.bam"), "wb").write(bgzf(bam_bytes(list(zip(names, lens)), ch)))
    json.dump({"genotype": [names[g] for g in L.true_genotype], "pairs": a.pairs}, open(os.path.join(a.root, "truth.json"), "w"))
    print(f"wrote {a.root}: {a.alleles} alleles, {a.pairs} read pairs, truth {L.true_genotype}")


if __name__ == "__main__":
    main()
