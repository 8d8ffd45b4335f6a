"""The readers and writers around recruitment (lcty_fastx.hip; src/seq/fastx.rs, src/seq/recruit.rs:1000-1030): host code, CPU tier.
Expected values are derived by hand from the cited lines (the reference has no tests)."""
import gzip

import numpy as np
import pytest

from locityper_amd import _lib, cdefs
from locityper_amd import io as lio


def bases_of(chunk, mate):
    ln, off = int(chunk.mate_len[mate]), int(chunk.mate_off[mate])
    idx = off + np.arange(ln)
    code = (chunk.bases2[idx >> 4] >> (2 * (idx & 15)).astype(np.uint32)) & 3
    isn = (chunk.nmask[idx >> 5] >> (idx & 31).astype(np.uint32)) & 1
    return "".join("N" if n else "ACGT"[c] for c, n in zip(code, isn))


def test_fastq_and_fasta_records_as_read_next_reads_them(tmp_path):
    p = tmp_path / "r.fq"
    # a description behind the name is dropped (fastx.rs:417-419); \\r\\n line ends (79-99); lower case and IUPAC are "not ACGT" (kmers.rs:178-190)
    p.write_text("@r1 first read\nACGTNacgtRT\n+\nIIIIIIIIIII\n@r2\r\nTTTT\r\n+r2\r\n!!!!\r\n")
    f = lio.Fastx(p)
    assert not f.paired
    ch = f.next(10)
    assert ch.n_pairs == 2 and list(ch.mate_len) == [11, 0, 4, 0] and list(ch.mate_off) == [0, 32, 32, 64, 64]
    assert bases_of(ch, 0) == "ACGTNNNNNNT" and bases_of(ch, 2) == "TTTT"
    assert f.next(10) is None
    w = lio.FastxWriters([tmp_path / "out0.fq", tmp_path / "out1.fq.gz"])
    f2 = lio.Fastx(p)
    f2.next(10)
    n = f2.write_recruited(w, [2, 1], [[0, 1], [1, 0]])
    w.close()
    assert n == 2
    assert (tmp_path / "out0.fq").read_text() == "@r1\nACGTNacgtRT\n+\nIIIIIIIIIII\n"            # write_fastq (fastx.rs:62-75): the name only
    assert gzip.open(tmp_path / "out1.fq.gz", "rt").read() == "@r1\nACGTNacgtRT\n+\nIIIIIIIIIII\n@r2\nTTTT\n+\n!!!!\n"
    fa = tmp_path / "r.fa.gz"
    with gzip.open(fa, "wt") as g:
        g.write(">s1 desc\nACGT\nAC\n>s2\nGG\n")                                               # a multi-line FASTA record (321-341)
    f3 = lio.Fastx(fa)
    ch = f3.next(1)
    assert ch.n_pairs == 1 and bases_of(ch, 0) == "ACGTAC"
    ch = f3.next(5)
    assert ch.n_pairs == 1 and bases_of(ch, 0) == "GG"
    w = lio.FastxWriters([tmp_path / "o.fa"])
    f3.write_recruited(w, [1], [[0]])                                                            # the records of the LAST chunk
    w.close()
    assert (tmp_path / "o.fa").read_text() == ">s2\nGG\n"                                       # write_fasta (46-57)
    assert f3.next(5) is None


def test_paired_input_two_files_and_interleaved(tmp_path):
    a, b, il = tmp_path / "a.fq", tmp_path / "b.fq", tmp_path / "il.fq"
    a.write_text("@p1/1\nACGT\n+\nIIII\n@p2/1\nCCCC\n+\nIIII\n")
    b.write_text("@p1/2\nTTGG\n+\nIIII\n@p2/2\nGGGA\n+\nIIII\n")
    il.write_text("@p1/1\nACGT\n+\nIIII\n@p1/2\nTTGG\n+\nIIII\n@p2/1\nCCCC\n+\nIIII\n@p2/2\nGGGA\n+\nIIII\n")
    for f in (lio.Fastx(a, b), lio.Fastx(il, interleaved=True)):
        assert f.paired
        ch = f.next(100)
        assert ch.n_pairs == 2 and list(ch.mate_len) == [4, 4, 4, 4] and [bases_of(ch, m) for m in range(4)] == ["ACGT", "TTGG", "CCCC", "GGGA"]
        w = lio.FastxWriters([tmp_path / "l0.fq"])
        f.write_recruited(w, [0, 1], [[0], [0]])
        w.close()
        # both mates of the recruited pair, one after the other (fastx.rs:141-150): what the mapper gets WITHOUT --interleaved
        assert (tmp_path / "l0.fq").read_text() == "@p2/1\nCCCC\n+\nIIII\n@p2/2\nGGGA\n+\nIIII\n"
    # chunks end at max_records and continue where they stopped
    f = lio.Fastx(a, b)
    assert f.next(1).n_pairs == 1 and f.next(1).n_pairs == 1 and f.next(1) is None


def test_reader_errors_are_the_references(tmp_path):
    def fails(text, what, **kw):
        p = tmp_path / "bad.fq"
        p.write_text(text)
        with pytest.raises(_lib.LocityperError) as e:
            f = lio.Fastx(p, **kw)
            while f.next(10) is not None:
                pass
        assert e.value.code == cdefs.ERR_INVALID_DATA and what in str(e.value), str(e.value)
    fails("@r1\nACGT\n", "Fastq record r1 is incomplete")                                       # fastx.rs:350-352
    fails("@r1\nACGT\nIIII\n@r2\n", "Fastq record r1 has incorrect format")                      # 353-354
    fails("@r1\nACGT\n+\nIII\n", "Fastq record r1 has non-matching sequence and qualities")      # 358-362
    fails(">s1\n", "Fasta record s1 has an empty sequence.")                                    # 327-330
    fails("@a/1\nAC\n+\nII\n@a/2\nAC\n+\nII\n@b/1\nAC\n+\nII\n", "Odd number of records in an interleaved input file(s)", interleaved=True)
    fails("@abcd\nAC\n+\nII\n@aXcd\nAC\n+\nII\n", "contains non matching first and second mate (abcd and aXcd)", interleaved=True)   # equal_names_fast looks at the third character from the end (101-109)
    a, b = tmp_path / "a.fq", tmp_path / "b.fq"
    a.write_text("@p1\nAC\n+\nII\n@p2\nAC\n+\nII\n")
    b.write_text("@p1\nAC\n+\nII\n")
    with pytest.raises(_lib.LocityperError) as e:
        f = lio.Fastx(a, b)
        while f.next(1) is not None:
            pass
    assert e.value.code == cdefs.ERR_INVALID_DATA and "Different number of records in paired-end input files" in str(e.value)
    with pytest.raises(_lib.LocityperError) as e:
        lio.Fastx(tmp_path / "reads.fq.lz4")
    assert e.value.code == cdefs.ERR_UNSUPPORTED
    with pytest.raises(_lib.LocityperError) as e:
        lio.Fastx(tmp_path / "missing.fq")
    assert e.value.code == cdefs.ERR_INVALID_INPUT
