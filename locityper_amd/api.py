"""Thin Python binding of the C ABI (include/locityper_hip.h), used by tests/ and bench.py.

Class and method names follow the reference interface of the path:
  Locus        ~ ContigSet + KmerCounts + ContigInfos + UniqueKmers (src/model/locs.rs:930-963, windows.rs:584-615)
  AllAlignments ~ model::locs::AllAlignments (load 1085-1185, best_aln_matrix 1203-1212)
  run_filter / truncate_ixs ~ src/solvers/solve.rs:52-122
Everything is computed by liblocityper_hip.so on the GPU; nothing here falls back to the CPU.
"""
import ctypes as C

import numpy as np

from . import cdefs
from ._lib import lib, check, VP, U32, U64, D
from .cdefs import Bg, Params, Solver, PAIR_ALN_DTYPE


def device_count():
    return int(lib().lcty_device_count())


def default_params():
    p = Params()
    lib().lcty_params_default(C.byref(p))
    return p


def resolve_params(p, bg):
    check(lib().lcty_params_resolve(C.byref(p), C.byref(bg)))
    return p


class _PinnedBlock:
    def __init__(self, addr):
        self.addr = addr

    def __del__(self):
        try:
            lib().lcty_host_free(self.addr)
        except Exception:
            pass


class Context:
    def __init__(self, device=0):
        self._h = VP()
        check(lib().lcty_ctx_create(device, C.byref(self._h)))

    def synchronize(self):
        check(lib().lcty_ctx_synchronize(self._h))

    def set_knob(self, name, value):
        """lcty_ctx_set_knob: a limit of the retry / batching machinery (tests lower them); value < 0 = default."""
        check(lib().lcty_ctx_set_knob(self._h, name.encode(), int(value)))

    def set_path(self, name, path):
        """lcty_ctx_set_path: a file the library is asked to write ("exact_dump"); None switches it off."""
        check(lib().lcty_ctx_set_path(self._h, name.encode(), None if path is None else str(path).encode()))

    def trim(self):
        """lcty_ctx_trim: release the solver workspaces kept between stages."""
        check(lib().lcty_ctx_trim(self._h))

    def pinned_like(self, arr):
        """A copy of `arr` in page-locked host memory (lcty_host_alloc): chunks built from such arrays upload at PCIe link rate.
        The memory is released when the returned array (and every view of it) is gone."""
        arr = np.ascontiguousarray(arr)
        p = VP()
        check(lib().lcty_host_alloc(self._h, max(arr.nbytes, 1), C.byref(p)))
        holder = _PinnedBlock(p.value)
        buf = (C.c_uint8 * max(arr.nbytes, 1)).from_address(p.value)
        out = np.frombuffer(buf, dtype=arr.dtype, count=arr.size).reshape(arr.shape)
        out[...] = arr
        buf._lcty_block = holder                  # the block lives as long as the buffer object the array is a view of
        return out

    def pinned_chunk(self, ch):
        """A ReadsChunk whose arrays lie in page-locked memory."""
        from .cdefs import ReadsChunk
        return ReadsChunk(*(self.pinned_like(a) for a in (ch.mate_len, ch.mate_off, ch.bases2, ch.nmask, ch.aln_off, ch.recs,
                                                          ch.cigar_off, ch.cigar)))

    def timing_reset(self):
        """Switches the HIP-event timing of this context on (it is off until the first call) and zeroes the totals."""
        check(lib().lcty_timing_reset(self._h))

    def timing(self, kernel):
        n, ms = U64(), D()
        check(lib().lcty_timing_get(self._h, kernel, C.byref(n), C.byref(ms)))
        return int(n.value), float(ms.value)

    def close(self):
        if self._h:
            lib().lcty_ctx_destroy(self._h)
            self._h = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


K_SCORE, K_PREFILTER, K_SOLVE, K_SOLVE_INIT, K_SOLVE_TABLE = 0, 1, 2, 3, 4
K_TRANSFER, K_ANNEAL = 5, 7
K_SOLVE_INIT_ANNEAL = 9          # the initialisation of an annealing stage's chains (K_SOLVE_INIT: a greedy or exact stage's)


class Locus:
    def __init__(self, ctx, seqs, seq_off, counts, cnt_off, k, bg, params):
        self.ctx = ctx
        self.seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        self.seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
        self.counts = np.ascontiguousarray(counts, dtype=np.uint16)
        self.cnt_off = np.ascontiguousarray(cnt_off, dtype=np.uint64)
        self.n_alleles = len(self.seq_off) - 1
        self.allele_len = np.diff(self.seq_off.astype(np.int64))
        self.k, self.bg, self.params = k, bg, params
        self._h = VP()
        check(lib().lcty_locus_create(ctx._h, self.n_alleles, self.seqs.ctypes.data, self.seq_off.ctypes.data,
                                      self.counts.ctypes.data, self.cnt_off.ctypes.data, k,
                                      C.byref(bg), C.byref(params), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().lcty_locus_destroy(self._h)
            self._h = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_hap_alns(self, entries, transfer_fails=3, max_div=0.05):
        """HapAlns (seq/transfer.rs:21-67): entries = [(id1 query, id2 target, raw CIGAR words, n_matches, aln_len)]."""
        n = len(entries)
        id1 = np.array([e[0] for e in entries], dtype=np.uint32)
        id2 = np.array([e[1] for e in entries], dtype=np.uint32)
        off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum([len(e[2]) for e in entries], out=off[1:])
        words = np.concatenate([np.asarray(e[2], dtype=np.uint32) for e in entries]) if n else np.zeros(1, dtype=np.uint32)
        nm = np.array([e[3] for e in entries], dtype=np.uint32)
        ln = np.array([e[4] for e in entries], dtype=np.uint32)
        import time
        t0 = time.perf_counter()
        check(lib().lcty_locus_set_hap_alns(self._h, n, id1.ctypes.data, id2.ctypes.data, off.ctypes.data, words.ctypes.data, nm.ctypes.data,
                                            ln.ctypes.data, transfer_fails, max_div))
        self.set_hap_alns_call_s = time.perf_counter() - t0                   # the library call alone (bench reports it)

    def n_unique_kmers(self):
        n = U64()
        check(lib().lcty_locus_n_unique_kmers(self._h, C.byref(n)))
        return int(n.value)

    def contig_info(self, a):
        ln = int(self.seq_off[a + 1] - self.seq_off[a])
        npos = ln - self.bg.neighb + 1
        gc = np.zeros(npos, dtype=np.uint8)
        uniq = np.zeros(npos, dtype=np.uint32)
        cc = np.zeros(npos, dtype=np.uint16)
        nw, rs = U32(), U32()
        check(lib().lcty_locus_contig_info(self._h, a, gc.ctypes.data, uniq.ctypes.data, cc.ctypes.data,
                                           C.byref(nw), C.byref(rs)))
        return gc, uniq, cc, nw.value, rs.value

    def edit_thresholds(self, read_len):
        g, p = U32(), U32()
        check(lib().lcty_locus_edit_thresholds(self._h, read_len, C.byref(g), C.byref(p)))
        return g.value, p.value

    def insert_lnprob(self, sizes):
        sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
        out = np.zeros(len(sizes), dtype=np.float64)
        pen = D()
        check(lib().lcty_locus_insert_lnprob(self._h, len(sizes), sizes.ctypes.data, out.ctypes.data, C.byref(pen)))
        return out, pen.value

    def set_explicit_weights(self, allele, start, end, value):
        """Explicit region weights (--reg-weights; load_explicit_weights, windows.rs:257-317) as parsed BED lines in file order."""
        al = np.ascontiguousarray(allele, dtype=np.uint32); st = np.ascontiguousarray(start, dtype=np.uint32)
        en = np.ascontiguousarray(end, dtype=np.uint32); va = np.ascontiguousarray(value, dtype=np.float64)
        if not (len(al) == len(st) == len(en) == len(va)):
            raise ValueError("explicit weights: columns of different lengths")
        check(lib().lcty_locus_set_explicit_weights(self._h, len(al), al.ctypes.data, st.ctypes.data, en.ctypes.data, va.ctypes.data))

    def window_weights(self):
        """ContigInfo::neighb_info weights of every moving-window position (alleles concatenated)."""
        n = sum(int(self.seq_off[a + 1] - self.seq_off[a]) - self.bg.neighb + 1 for a in range(self.n_alleles))
        out = np.zeros(n, dtype=np.float64)
        check(lib().lcty_locus_window_weights(self._h, out.ctypes.data))
        return out

    def depth_table(self, width):
        """lcty_locus_depth_table: the extended depth table of the solver stages, [101][width rounded up to a power of two]."""
        w = U32(width)
        check(lib().lcty_locus_depth_table(self._h, C.byref(w), None))
        out = np.zeros((cdefs.GC_BINS, int(w.value)), dtype=np.float64)
        check(lib().lcty_locus_depth_table(self._h, C.byref(w), out.ctypes.data))
        return out

    def depth_lut(self):
        out = np.zeros((cdefs.GC_BINS, cdefs.DEPTH_CACHE), dtype=np.float64)
        check(lib().lcty_locus_depth_lut(self._h, out.ctypes.data))
        return out


class AllAlignments:
    """Device-resident batch of read pairs and the products of AllAlignments::load."""

    def __init__(self, locus, cap_pairs, cap_bases, cap_recs, cap_cigar, streaming_chunk_pairs=None, cap_pair_alns=0):
        """streaming_chunk_pairs: a streaming batch (lcty_reads_create_streaming) — cap_bases / cap_recs / cap_cigar are then the
        capacities of ONE chunk, cap_pairs that of the whole batch; append + score chunk after chunk."""
        self.locus = locus
        self._h = VP()
        if streaming_chunk_pairs is None:
            check(lib().lcty_reads_create(locus._h, cap_pairs, cap_bases, cap_recs, cap_cigar, C.byref(self._h)))
        else:
            check(lib().lcty_reads_create_streaming(locus._h, cap_pairs, streaming_chunk_pairs, cap_bases, cap_recs, cap_cigar,
                                                    cap_pair_alns, C.byref(self._h)))
        self._scored = False

    @classmethod
    def load_streaming(cls, locus, chunks, cap_pair_alns=0):
        """AllAlignments::load of a batch whose records do not have to fit the device together: one chunk resident at a time."""
        chunks = list(chunks)
        self = cls(locus, sum(c.n_pairs for c in chunks), (max(c.n_bases for c in chunks) + 31) // 32 * 32,
                   max(len(c.recs) for c in chunks), max(len(c.cigar) for c in chunks),
                   streaming_chunk_pairs=max(c.n_pairs for c in chunks), cap_pair_alns=cap_pair_alns)
        for c in chunks:
            self.append(c)
            self.score()
        return self

    @classmethod
    def load(cls, locus, chunks, counted=False):
        """AllAlignments::load: upload `chunks` (ReadsChunk or list of them) and score them. counted: hand the records over as
        lcty_aln_counted entries (operations counted here, on the host, as the caller of lcty_reads_append_counted would)."""
        if not isinstance(chunks, (list, tuple)):
            chunks = [chunks]
        self = cls(locus, sum(c.n_pairs for c in chunks), sum(c.n_bases for c in chunks),
                   sum(len(c.recs) for c in chunks), 0 if counted else sum(len(c.cigar) for c in chunks))
        for c in chunks:
            self.append(c, counted=counted)
        self.score()
        return self

    def reset(self, locus):
        """lcty_reads_reset: an empty batch again, bound to `locus` (buffers stay)."""
        check(lib().lcty_reads_reset(self._h, locus._h))
        self.locus = locus
        self._scored = False

    def append(self, chunk, counted=False):
        """counted: True (the operations of the chunk's records are counted here, on the host) or the (n_records, 4) u32 array of
        lcty_aln_counted entries made earlier (ReadsChunk.counted), e.g. in page-locked memory"""
        hs = chunk.host_struct()
        if counted is not False and counted is not None:
            alns = np.ascontiguousarray(chunk.counted(self.locus.allele_len) if counted is True else counted, dtype=np.uint32)
            check(lib().lcty_reads_append_counted(self._h, C.byref(hs), alns.ctypes.data))
        else:
            check(lib().lcty_reads_append(self._h, C.byref(hs)))
        self._scored = False

    def score(self):
        check(lib().lcty_score_reads(self._h))
        self._scored = True

    def recover(self):
        """Alignment recovery (transfer.rs:70-140) between two scoring passes; returns the number of transferred alignments."""
        n = U64()
        check(lib().lcty_recover_alignments(self._h, C.byref(n)))
        if n.value: self.score()                 # nothing transferred: the batch is as it was scored
        return int(n.value)

    def recover_dp_cells(self):
        """Cells of the aligner's matrices filled by the last recover()."""
        n = U64()
        check(lib().lcty_recover_dp_cells(self._h, C.byref(n)))
        return int(n.value)

    def recover_stats(self):
        """Read pairs the last recover() took at each of the three lane-scratch levels."""
        out = (C.c_uint64 * 3)()
        check(lib().lcty_recover_stats(self._h, out))
        return [int(x) for x in out]

    def close(self):
        if self._h:
            lib().lcty_reads_destroy(self._h)
            self._h = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_pairs(self):
        n = U64()
        check(lib().lcty_reads_n_pairs(self._h, C.byref(n)))
        return int(n.value)

    def n_good(self):
        n = U64()
        check(lib().lcty_reads_n_good(self._h, C.byref(n)))
        return int(n.value)

    def status(self):
        n = self.n_pairs
        status = np.zeros(n, dtype=np.uint8)
        weight = np.zeros(n, dtype=np.float64)
        unm = np.zeros(n, dtype=np.float64)
        uniq = np.zeros(2 * n, dtype=np.uint16)
        check(lib().lcty_reads_get_status(self._h, status.ctypes.data, weight.ctypes.data, unm.ctypes.data, uniq.ctypes.data))
        return status, weight, unm, uniq

    def best_aln_matrix(self):
        """[A][n_good] f64 (locs.rs:1203-1212)."""
        ng = self.n_good()
        out = np.zeros((self.locus.n_alleles, ng), dtype=np.float64)
        check(lib().lcty_best_aln_matrix(self._h, out.ctypes.data))
        return out

    def pair_alns(self):
        n = self.n_pairs
        off = np.zeros(n + 1, dtype=np.uint64)
        check(lib().lcty_reads_get_pair_alns(self._h, off.ctypes.data, None, 0))
        out = np.zeros(int(off[-1]), dtype=PAIR_ALN_DTYPE)
        check(lib().lcty_reads_get_pair_alns(self._h, off.ctypes.data, out.ctypes.data, len(out)))
        return off, out

    # ---- prefilter ----
    def run_filter(self, genotypes=None, priors=None, ploidy=2):
        """run_filter scores (solve.rs:101-119). genotypes None = all multisets of `ploidy`."""
        A = self.locus.n_alleles
        if genotypes is None:
            n = count_genotypes(A, ploidy)
            gptr = None
        else:
            genotypes = np.ascontiguousarray(genotypes, dtype=np.uint16)
            n, ploidy = genotypes.shape
            gptr = genotypes.ctypes.data
        scores = np.zeros(n, dtype=np.float64)
        pr = None if priors is None else np.ascontiguousarray(priors, dtype=np.float64)
        check(lib().lcty_prefilter(self._h, gptr, n, ploidy, None if pr is None else pr.ctypes.data, scores.ctypes.data))
        return scores

    def records(self):
        """lcty_reads_get_records: (aln_off, recs, cigar_off, cigar) as the batch holds them now (after recover(): with the transferred
        alignments)."""
        n = self.n_pairs
        aln_off = np.zeros(n + 1, dtype=np.uint64); cig_off = np.zeros(n + 1, dtype=np.uint64)
        check(lib().lcty_reads_get_records(self._h, aln_off.ctypes.data, None, 0, cig_off.ctypes.data, None, 0))
        recs = np.zeros(max(int(aln_off[-1]), 1), dtype=cdefs.ALN_REC_DTYPE); cigar = np.zeros(max(int(cig_off[-1]), 1), dtype=np.uint32)
        check(lib().lcty_reads_get_records(self._h, aln_off.ctypes.data, recs.ctypes.data, len(recs), cig_off.ctypes.data, cigar.ctypes.data, len(cigar)))
        return aln_off, recs[:int(aln_off[-1])], cig_off, cigar[:int(cig_off[-1])]

    def prefilter_async(self, ploidy=2):
        check(lib().lcty_prefilter_async(self._h, ploidy))

    def prefilter_scores(self):
        n = count_genotypes(self.locus.n_alleles, 2)
        scores = np.zeros(n, dtype=np.float64)
        check(lib().lcty_prefilter_scores(self._h, scores.ctypes.data, n))
        return scores

    def prefilter_add_priors(self, priors):
        priors = np.ascontiguousarray(priors, dtype=np.float64)
        check(lib().lcty_prefilter_add_priors(self._h, priors.ctypes.data, len(priors)))

    def prefilter_truncate(self, filt_diff, min_size, threads):
        """lcty_prefilter_truncate: truncate_ixs on the scores the last prefilter call left on the device; the kept genotype indices
        sorted by (score desc, index asc)."""
        keep = U64()
        ixs = np.empty(count_genotypes(self.locus.n_alleles, 2), dtype=np.uint64)      # room for all of them: one call, one sort
        check(lib().lcty_prefilter_truncate(self._h, filt_diff, min_size, threads, ixs.ctypes.data, len(ixs), C.byref(keep)))
        return ixs[:int(keep.value)].copy()


def default_solver(kind):
    s = Solver()
    check(lib().lcty_solver_default(C.byref(s), kind))
    return s


def chain_seeds(master_seed, n):
    out = np.zeros(n, dtype=np.uint64)
    check(lib().lcty_chain_seeds(master_seed, n, out.ctypes.data))
    return out


def solve_stage(aa, genotypes, solver, attempts, seeds, priors=None):
    """One solver stage (solve.rs:816-843) on the GPU: returns (lik_mean, lik_var, liks[n_gt][attempts])."""
    genotypes = np.ascontiguousarray(genotypes, dtype=np.uint16)
    n, ploidy = genotypes.shape
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    assert len(seeds) == n * attempts
    mean, var, liks = np.zeros(n), np.zeros(n), np.zeros((n, attempts))
    pri = None if priors is None else np.ascontiguousarray(priors, dtype=np.float64)
    check(lib().lcty_solve_stage(aa._h, genotypes.ctypes.data, n, ploidy, None if pri is None else pri.ctypes.data,
                                 C.byref(solver), attempts, seeds.ctypes.data, mean.ctypes.data, var.ctypes.data,
                                 liks.ctypes.data))
    return mean, var, liks


def rng_seed_from_u64(seed):
    """The four words of XoshiroRng::seed_from_u64(seed) (ext/rand.rs:3-22)."""
    st = np.zeros(4, dtype=np.uint64)
    check(lib().lcty_rng_seed_from_u64(seed, st.ctypes.data))
    return st


def rng_next_u64(state):
    """next_u64 of the xoshiro256++ generator whose four words `state` holds (advanced in place)."""
    out = U64(0)
    check(lib().lcty_rng_next_u64(state.ctypes.data, C.byref(out)))
    return int(out.value)


def solve_given(locus, read_ixs, ln_prob, windows, window_gc, window_weight, depth_contrib, aln_contrib, solver, rng_state, wshifts=None,
                tables=None, tables_id=0, deepest_only=False):
    """`Solver::solve` (solvers/mod.rs:59-72) on a GenotypeAlignments handed over as arrays (lcty_solve_given): one chain on the device
    over exactly these locations and window distributions. Returns (likelihood, read_assgn u16[n_reads], (aln_lik, depth_lik));
    `rng_state` (four uint64 words, xoshiro256++) is advanced in place by the one draw the call takes. With `tables` (f64[n_rows][width])
    `locus` is a Context and window_gc names rows of the caller's own distributions (lcty_solve_given_tables)."""
    read_ixs = np.ascontiguousarray(read_ixs, dtype=np.uint64)
    ln_prob = np.ascontiguousarray(ln_prob, dtype=np.float64)
    windows = np.ascontiguousarray(windows, dtype=np.uint32).reshape(-1)
    window_gc = np.ascontiguousarray(window_gc, dtype=np.uint8)
    window_weight = np.ascontiguousarray(window_weight, dtype=np.float64)
    n_reads = len(read_ixs) - 1
    v = cdefs.GtAlnsView()
    v.n_reads = n_reads
    v.read_ixs, v.ln_prob, v.windows = read_ixs.ctypes.data, ln_prob.ctypes.data, windows.ctypes.data
    v.n_windows = len(window_weight)
    v.window_gc, v.window_weight = window_gc.ctypes.data, window_weight.ctypes.data
    ws = None
    if wshifts is not None:
        ws = np.ascontiguousarray(wshifts, dtype=np.uint32)
        v.n_contigs, v.wshifts = len(ws) - 1, ws.ctypes.data
    v.depth_contrib, v.aln_contrib = depth_contrib, aln_contrib
    assgn = np.zeros(max(n_reads, 1), dtype=np.uint16)
    parts = np.zeros(2, dtype=np.float64)
    lik = D(0.0)
    if deepest_only:
        d = U32(0)
        check(lib().lcty_gt_alns_deepest(C.byref(v), C.byref(d)))
        return int(d.value)
    rs = None if rng_state is None else rng_state.ctypes.data
    if tables is None:
        check(lib().lcty_solve_given(locus._h, C.byref(v), C.byref(solver), rs, assgn.ctypes.data, parts.ctypes.data, C.byref(lik)))
    else:
        tables = np.ascontiguousarray(tables, dtype=np.float64)
        t = cdefs.DepthTables(tables.shape[0], tables.shape[1], tables.ctypes.data, tables_id)
        check(lib().lcty_solve_given_tables(locus._h, C.byref(v), C.byref(t), C.byref(solver), rs, assgn.ctypes.data, parts.ctypes.data,
                                            C.byref(lik)))
    return float(lik.value), assgn[:n_reads], parts


def solve_stage_from_shards(shards, genotypes, solver, attempts, seeds, priors=None):
    """One solver stage over the reads of several batches of one locus on one device (lcty_solve_stage_from_shards): `shards` in
    read order; equals solve_stage on the unsharded batch."""
    genotypes = np.ascontiguousarray(genotypes, dtype=np.uint16)
    n, ploidy = genotypes.shape
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    assert len(seeds) == n * attempts
    mean, var, liks = np.zeros(n), np.zeros(n), np.zeros((n, attempts))
    pri = None if priors is None else np.ascontiguousarray(priors, dtype=np.float64)
    handles = (VP * len(shards))(*[s._h for s in shards])
    check(lib().lcty_solve_stage_from_shards(handles, len(shards), genotypes.ctypes.data, n, ploidy,
                                             None if pri is None else pri.ctypes.data, C.byref(solver), attempts, seeds.ctypes.data,
                                             mean.ctypes.data, var.ctypes.data, liks.ctypes.data))
    return mean, var, liks


def assignment_counts(aa, genotype, solver, attempts, seeds):
    """Per-read assignment counts of one genotype (update_counts, assgn.rs:374-378): (read_off[n_good+1], counts u16)."""
    genotype = np.ascontiguousarray(genotype, dtype=np.uint16).reshape(-1)
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    assert len(seeds) == attempts
    n_good = aa.n_good()
    off = np.zeros(n_good + 1, dtype=np.uint64)
    n = U64()
    args = (aa._h, genotype.ctypes.data, len(genotype), C.byref(solver), attempts, seeds.ctypes.data, off.ctypes.data)
    check(lib().lcty_assignment_counts(*args, None, 0, C.byref(n)))
    counts = np.zeros(int(n.value), dtype=np.uint16)
    check(lib().lcty_assignment_counts(*args, counts.ctypes.data, len(counts), C.byref(n)))
    return off, counts


def count_unexplained(aa, genotype):
    """Genotyping::count_unexplained_reads (solve.rs:718-729)."""
    genotype = np.ascontiguousarray(genotype, dtype=np.uint16).reshape(-1)
    out = U32()
    check(lib().lcty_count_unexplained(aa._h, genotype.ctypes.data, len(genotype), C.byref(out)))
    return int(out.value)


def call_checks(genotypes, ln_probs, n_reads, dist=None, n_alleles=0):
    """find_weighted_dist / check_first_prob / check_num_of_reads (solve.rs:621-675): (distances, weighted_dist, warnings)."""
    genotypes = np.ascontiguousarray(genotypes, dtype=np.uint16)
    n, ploidy = genotypes.shape
    lp = np.ascontiguousarray(ln_probs, dtype=np.float64)
    dm = None if dist is None else np.ascontiguousarray(dist, dtype=np.uint32)
    out = np.zeros(n, dtype=np.uint32)
    wd, warn = D(), U32()
    check(lib().lcty_call_checks(genotypes.ctypes.data, n, ploidy, lp.ctypes.data, n_reads, None if dm is None else dm.ctypes.data,
                                 n_alleles if dm is None else dm.shape[0], out.ctypes.data, C.byref(wd), C.byref(warn)))
    return out, float(wd.value), int(warn.value)


def solve_stats(aa):
    """(chains, iterations, accepted moves) of the last solve_stage on this batch."""
    c, i, a = U64(), U64(), U64()
    check(lib().lcty_solve_stats(aa._h, C.byref(c), C.byref(i), C.byref(a)))
    return int(c.value), int(i.value), int(a.value)


def discard_improbable(lik_mean, lik_var, attempts, ixs, prob_thresh, out_size, threads):
    lik_mean = np.ascontiguousarray(lik_mean, dtype=np.float64)
    lik_var = np.ascontiguousarray(lik_var, dtype=np.float64)
    attempts = np.ascontiguousarray(attempts, dtype=np.uint32)
    ixs = np.ascontiguousarray(ixs, dtype=np.uint64).copy()
    keep = U64()
    check(lib().lcty_discard_improbable(lik_mean.ctypes.data, lik_var.ctypes.data, attempts.ctypes.data, ixs.ctypes.data,
                                        len(ixs), prob_thresh, out_size, threads, C.byref(keep)))
    return ixs[:int(keep.value)]


def produce_result(lik_mean, lik_var, attempts, ixs, prob_thresh, out_bams=0):
    lik_mean = np.ascontiguousarray(lik_mean, dtype=np.float64)
    lik_var = np.ascontiguousarray(lik_var, dtype=np.float64)
    attempts = np.ascontiguousarray(attempts, dtype=np.uint32)
    ixs = np.ascontiguousarray(ixs, dtype=np.uint64)
    out_ixs, out_lp = np.zeros(50, dtype=np.uint64), np.zeros(50, dtype=np.float64)
    n, q = U64(), D()
    check(lib().lcty_produce_result(lik_mean.ctypes.data, lik_var.ctypes.data, attempts.ctypes.data, ixs.ctypes.data, len(ixs),
                                    prob_thresh, out_bams, out_ixs.ctypes.data, out_lp.ctypes.data, C.byref(n), C.byref(q)))
    return out_ixs[:int(n.value)], out_lp[:int(n.value)], q.value


def count_genotypes(n_alleles, ploidy):
    return int(lib().lcty_count_genotypes(n_alleles, ploidy))


def generate_genotypes(n_alleles, ploidy):
    n = count_genotypes(n_alleles, ploidy)
    out = np.zeros((n, ploidy), dtype=np.uint16)
    check(lib().lcty_generate_genotypes(n_alleles, ploidy, out.ctypes.data, n))
    return out


def parse_kmer_counts(data):
    """KmerCounts::load (counts.rs:127-150) of the decompressed `kmers.bin`: (k, cnt_off[n_contigs+1], counts u16, bytes consumed)
    of the first block (the off-target counts)."""
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    k, n, used = U32(), U32(), U64()
    check(lib().lcty_kmer_counts_parse(buf.ctypes.data, len(buf), C.byref(k), C.byref(n), None, 0, None, 0, C.byref(used)))
    off = np.zeros(n.value + 1, dtype=np.uint64)
    counts = np.zeros(max(int(used.value), 1), dtype=np.uint16)
    check(lib().lcty_kmer_counts_parse(buf.ctypes.data, len(buf), C.byref(k), C.byref(n), off.ctypes.data, n.value, counts.ctypes.data,
                                       len(counts), C.byref(used)))
    return int(k.value), off, counts[:int(off[-1])], int(used.value)


def truncate_ixs(scores, ixs, filt_diff, min_size, threads):
    """truncate_ixs (solve.rs:52-84): returns the kept indices sorted by (score desc, index asc)."""
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    ixs = np.ascontiguousarray(ixs, dtype=np.uint64).copy()
    keep = U64()
    check(lib().lcty_truncate(scores.ctypes.data, ixs.ctypes.data, len(ixs), filt_diff, min_size, threads, C.byref(keep)))
    return ixs[:int(keep.value)]


# ---------------------------------------------------------------- solve::solve (src/solvers/solve.rs:926-981)
DEFAULT_SCHEME = (("greedy", 5000, 1), ("anneal", 20, 20))      # "-S greedy:i=5k,a=1 -S anneal:i=20,a=20" (solve.rs:211-230)


def solve(aa, params, scheme=DEFAULT_SCHEME, master_seed=1, priors=None, ploidy=2, genotypes=None, solvers=None):
    """The genotyping of one locus after AllAlignments::load, following solve::solve + solve_single_thread:
    run_filter/truncate_ixs when there are more genotypes than the first stage takes, then every stage
    (skipped when the survivors already fit the next stage's input, solve.rs:805-809) with
    discard_improbable_genotypes in between, and produce_result at the end.

    Returns dict(genotypes, ln_probs, quality, lik_mean, lik_var, attempts, kept_per_stage)."""
    A = aa.locus.n_alleles
    gts = generate_genotypes(A, ploidy) if genotypes is None else np.ascontiguousarray(genotypes, dtype=np.uint16)
    n = len(gts)
    pri = np.zeros(n) if priors is None else np.ascontiguousarray(priors, dtype=np.float64)
    ixs = np.arange(n, dtype=np.uint64)
    threads = max(1, int(params.threads))     # run_filter gets data.threads (solve.rs:945); discard gets it too unless it is 1 (797, 1087)
    kept = []
    if params.dont_skip or scheme[0][1] < n:
        if genotypes is None:
            scores = aa.run_filter(ploidy=ploidy)
            scores = scores + pri
        else:
            scores = aa.run_filter(gts, pri)
        ixs = truncate_ixs(scores, ixs, params.filt_diff, scheme[0][1], threads)
        kept.append(len(ixs))
    lik_mean = np.full(n, np.nan)
    lik_var = np.full(n, np.nan)
    att = np.zeros(n, dtype=np.uint32)
    seed_base = 0
    all_seeds = None
    for si, (name, in_size, attempts) in enumerate(scheme):
        out_size = scheme[si + 1][1] if si + 1 < len(scheme) else None
        if not (params.dont_skip or out_size is None or out_size < len(ixs)):
            continue                                               # "Skipping stage, not enough genotypes"
        solver = solvers[si] if solvers else default_solver(cdefs.SOLVER_GREEDY if name == "greedy" else cdefs.SOLVER_ANNEAL)
        seeds = chain_seeds(master_seed + 0x9E3779B97F4A7C15 * (si + 1) & 0xFFFFFFFFFFFFFFFF, len(ixs) * attempts)
        m, v, _ = solve_stage(aa, gts[ixs], solver, attempts, seeds, priors=pri[ixs])
        lik_mean[ixs], lik_var[ixs], att[ixs] = m, v, attempts
        if out_size is not None:
            ixs = discard_improbable(lik_mean, lik_var, att, ixs, params.prob_thresh, out_size, threads)
        kept.append(len(ixs))
    out_ixs, ln_probs, quality = produce_result(lik_mean, lik_var, att, ixs, params.prob_thresh)
    unexpl = count_unexplained(aa, gts[out_ixs[0]])
    return dict(unexpl_reads=unexpl, genotypes=gts[out_ixs], ixs=out_ixs, ln_probs=ln_probs, quality=quality, lik_mean=lik_mean, lik_var=lik_var,
                attempts=att, kept_per_stage=kept)


def default_stages():
    """Scheme::default (solve.rs:211-230) as an array of lcty_stage."""
    st = (cdefs.Stage * 2)()
    n = U32()
    check(lib().lcty_stages_default(st, C.byref(n)))
    return st


def solve_locus(aa, stages=None, master_seed=1, priors=None, ploidy=2):
    """lcty_solve: the library's own solve::solve (scheme loop in C++). Returns (Call, lik_mean, lik_var, attempts)."""
    stages = default_stages() if stages is None else stages
    G = count_genotypes(aa.locus.n_alleles, ploidy)
    pri = None if priors is None else np.ascontiguousarray(priors, dtype=np.float64)
    mean, var, att = np.zeros(G), np.zeros(G), np.zeros(G, dtype=np.uint32)
    call = cdefs.Call()
    check(lib().lcty_solve(aa._h, ploidy, stages, len(stages), master_seed, None if pri is None else pri.ctypes.data, C.byref(call),
                           mean.ctypes.data, var.ctypes.data, att.ctypes.data))
    return call, mean, var, att


def solve_queue(batches, stages=None, master_seeds=None, priors=None, ploidy=2):
    """lcty_solve_queue: score + solve every batch of the list (loci of one context), the last stage of each overlapped with the
    next entry. Returns the list of Call structs, one per entry."""
    stages = default_stages() if stages is None else stages
    n = len(batches)
    handles = (VP * n)(*[b._h for b in batches])
    seeds = np.ascontiguousarray(np.arange(1, n + 1) if master_seeds is None else master_seeds, dtype=np.uint64)
    assert len(seeds) == n
    pri_ptr = None
    keep = []
    if priors is not None:
        arr = (VP * n)()
        for i, p in enumerate(priors):
            if p is not None:
                keep.append(np.ascontiguousarray(p, dtype=np.float64))
                arr[i] = keep[-1].ctypes.data
        pri_ptr = arr
    calls = (cdefs.Call * n)()
    check(lib().lcty_solve_queue(handles, n, ploidy, stages, len(stages), seeds.ctypes.data, pri_ptr, calls))
    return list(calls)


def solve_queue_fed(n_loci, acquire, release=None, stages=None, master_seeds=None, ploidy=2):
    """lcty_solve_queue_fed: the queue with its batches handed over one at a time. acquire(i) -> AllAlignments (filled; may block until a
    loader thread is done with it), release(i) is called when nothing of position i's batch is in use any more. The library call
    runs without the interpreter lock; the callbacks take it."""
    stages = default_stages() if stages is None else stages
    seeds = np.ascontiguousarray(np.arange(1, n_loci + 1) if master_seeds is None else master_seeds, dtype=np.uint64)
    assert len(seeds) == n_loci
    failure = []

    @C.CFUNCTYPE(VP, VP, U32)
    def _acquire(_user, i):
        try:
            b = acquire(int(i))
            return None if b is None else b._h.value
        except BaseException as e:               # an exception cannot cross the C frames: the queue ends, the exception is raised below
            failure.append(e)
            return None

    @C.CFUNCTYPE(None, VP, U32)
    def _release(_user, i):
        try:
            if release is not None:
                release(int(i))
        except BaseException as e:
            failure.append(e)

    calls = (cdefs.Call * n_loci)()
    rc = lib().lcty_solve_queue_fed(n_loci, C.cast(_acquire, VP), C.cast(_release, VP), None, ploidy, stages, len(stages), seeds.ctypes.data, None, calls)
    if failure:
        raise failure[0]
    check(rc)
    return list(calls)


K_RECRUIT = 6
K_MAP = 8


def map_params(long_reads=False, **over):
    """lcty_map_params_default (seed length 15, a seed every 5 bases, scores 2 / 8 / end bonus 10, secondary records from score 50) or,
    with long_reads, lcty_map_params_default_long (a seed every 16 bases, scores 2 / 4 / gaps 4 + 2 n, every record kept)."""
    p = cdefs.MapParams()
    check((lib().lcty_map_params_default_long if long_reads else lib().lcty_map_params_default)(C.byref(p)))
    for k, v in over.items():
        setattr(p, k, v)
    return p


def build_map_index(locus, basis, k=15):
    """lcty_locus_build_map_index: the k-mers of the basis alleles of the locus."""
    basis = np.ascontiguousarray(basis, dtype=np.uint16)
    check(lib().lcty_locus_build_map_index(locus._h, basis.ctypes.data, len(basis), k))


def map_append(aa, chunk, params):
    """lcty_reads_map_append: the read ends of `chunk` mapped onto the basis alleles, the records straight into the batch `aa`."""
    h = chunk.host_struct()
    check(lib().lcty_reads_map_append(aa._h, C.byref(h), C.byref(params)))
    aa._scored = False


def map_reads(locus, chunk, params):
    """lcty_map_reads: the read ends of `chunk` (its sequence fields) onto the basis alleles; returns a ReadsChunk with the found
    records, =/X/S CIGARs and the bases in BAM orientation — ready for AllAlignments.append."""
    h = chunk.host_struct()
    n = chunk.n_pairs
    aln_off = np.zeros(n + 1, dtype=np.uint64); cig_off = np.zeros(n + 1, dtype=np.uint64)
    check(lib().lcty_map_reads(locus._h, C.byref(h), C.byref(params), aln_off.ctypes.data, None, 0, cig_off.ctypes.data, None, 0, None, None))
    recs = np.zeros(int(aln_off[-1]), dtype=cdefs.ALN_REC_DTYPE)
    cigar = np.zeros(max(int(cig_off[-1]), 1), dtype=np.uint32)
    b2 = np.zeros_like(chunk.bases2); nm = np.zeros_like(chunk.nmask)
    check(lib().lcty_map_reads(locus._h, C.byref(h), C.byref(params), aln_off.ctypes.data, recs.ctypes.data, len(recs), cig_off.ctypes.data,
                               cigar.ctypes.data, len(cigar), b2.ctypes.data, nm.ctypes.data))
    return cdefs.ReadsChunk(chunk.mate_len, chunk.mate_off, b2, nm, aln_off, recs, cig_off, cigar[:int(cig_off[-1])])


def recruit_params(technology=cdefs.TECH_ILLUMINA, paired=True, **over):
    """recruit::Params defaults (DEFAULT_MINIM_KW, match length 2000, k-mer threshold 50, Technology::default_match_frac)."""
    p = cdefs.RecruitParams()
    check(lib().lcty_recruit_params_default(C.byref(p), technology, int(paired)))
    for k, v in over.items():
        setattr(p, k, v)
    return p


class Targets:
    """recruit::Targets (TargetBuilder::add per locus, then finalize) and Targets::recruit_* on the device."""

    def __init__(self, ctx, params):
        self.ctx = ctx
        self._h = VP()
        check(lib().lcty_targets_create(ctx._h, C.byref(params), C.byref(self._h)))
        self.n_loci = 0

    def add_locus(self, seqs, seq_off, counts, cnt_off, base_k):
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8); seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
        counts = np.ascontiguousarray(counts, dtype=np.uint16); cnt_off = np.ascontiguousarray(cnt_off, dtype=np.uint64)
        ix = U32()
        check(lib().lcty_targets_add_locus(self._h, len(seq_off) - 1, seqs.ctypes.data, seq_off.ctypes.data, counts.ctypes.data,
                                           cnt_off.ctypes.data, base_k, C.byref(ix)))
        self.n_loci += 1
        return int(ix.value)

    def finalize(self):
        n = U64()
        check(lib().lcty_targets_finalize(self._h, C.byref(n)))
        return int(n.value)

    def recruit(self, chunk, paired=True, max_out=8):
        """Loci of every read pair (single read) of the chunk: list of sorted lists."""
        hs = chunk.host_struct()
        n = chunk.n_pairs
        cnt = np.zeros(max(n, 1), dtype=np.uint32); loci = np.zeros(max(n, 1) * max_out, dtype=np.uint32)
        check(lib().lcty_recruit(self._h, C.byref(hs), int(paired), max_out, cnt.ctypes.data, loci.ctypes.data))
        return cnt[:n], loci.reshape(-1, max_out)[:n]

    def close(self):
        if self._h:
            lib().lcty_targets_destroy(self._h)
            self._h = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


COMM_ID_BYTES = 128


def comm_unique_id():
    """ncclGetUniqueId (rank 0); hand the bytes to the other ranks (locityper_amd.dist.make_comm does that over torch.distributed)."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    check(lib().lcty_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """RCCL communicator of one process per GPU (lcty_comm): the SUM all-reduce of read-sharded run_filter scores."""

    def __init__(self, ctx, n_ranks, rank, unique_id):
        assert len(unique_id) == COMM_ID_BYTES
        self.ctx = ctx
        self._h = VP()
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        check(lib().lcty_comm_create(ctx._h, n_ranks, rank, buf, C.byref(self._h)))
        self.n_ranks, self.rank = n_ranks, rank

    def rccl_ranks(self):
        """(ranks, rank) as RCCL reports them for this communicator (ncclCommCount / ncclCommUserRank)."""
        n, r = C.c_int32(0), C.c_int32(0)
        check(lib().lcty_comm_ranks(self._h, C.byref(n), C.byref(r)))
        return int(n.value), int(r.value)

    def prefilter_allreduce(self, aa):
        check(lib().lcty_prefilter_allreduce(aa._h, self._h))

    def solve_stage(self, aa, genotypes, solver, attempts, seeds, priors=None):
        """solve_stage with the chains dealt to the ranks (lcty_solve_stage_sharded): same arguments on every rank, the results
        of all genotypes on every rank."""
        genotypes = np.ascontiguousarray(genotypes, dtype=np.uint16)
        n, ploidy = genotypes.shape
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        assert len(seeds) == n * attempts
        mean, var, liks = np.zeros(n), np.zeros(n), np.zeros((n, attempts))
        pri = None if priors is None else np.ascontiguousarray(priors, dtype=np.float64)
        check(lib().lcty_solve_stage_sharded(aa._h, self._h, genotypes.ctypes.data, n, ploidy, None if pri is None else pri.ctypes.data,
                                             C.byref(solver), attempts, seeds.ctypes.data, mean.ctypes.data, var.ctypes.data,
                                             liks.ctypes.data))
        return mean, var, liks

    def solve_stage_read_sharded(self, shard, genotypes, solver, attempts, seeds, priors=None):
        """One solver stage of a locus whose reads are sharded over the ranks (lcty_solve_stage_read_sharded): `shard` holds this
        rank's block of the read list; same stage arguments on every rank, the results of all genotypes on every rank."""
        genotypes = np.ascontiguousarray(genotypes, dtype=np.uint16)
        n, ploidy = genotypes.shape
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        assert len(seeds) == n * attempts
        mean, var, liks = np.zeros(n), np.zeros(n), np.zeros((n, attempts))
        pri = None if priors is None else np.ascontiguousarray(priors, dtype=np.float64)
        check(lib().lcty_solve_stage_read_sharded(shard._h, self._h, genotypes.ctypes.data, n, ploidy,
                                                  None if pri is None else pri.ctypes.data, C.byref(solver), attempts,
                                                  seeds.ctypes.data, mean.ctypes.data, var.ctypes.data, liks.ctypes.data))
        return mean, var, liks

    def close(self):
        if self._h:
            lib().lcty_comm_destroy(self._h)
            self._h = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def counts_to_posteriors(counts, attempts):
    """count_to_prob (model/bam.rs:56-67) for every entry of an assignment-count array: (probability f32, MAPQ u8)."""
    counts = np.ascontiguousarray(counts, dtype=np.uint16)
    prob = np.zeros(len(counts), dtype=np.float32); mapq = np.zeros(len(counts), dtype=np.uint8)
    check(lib().lcty_counts_to_posteriors(counts.ctypes.data, len(counts), attempts, prob.ctypes.data, mapq.ctypes.data))
    return prob, mapq
