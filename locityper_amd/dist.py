"""Multi-GPU plumbing: one process per GPU (torch.distributed launcher / env contract).

The path shards at three levels:
  * loci are independent (command/genotype.rs:1331-1351): round-robin over ranks, no collective;
  * inside one locus reads contribute additively to run_filter scores (solvers/solve.rs:105-119):
    read shards -> one SUM all-reduce of the G-long f64 score vector — on the devices through RCCL (make_comm +
    Comm.prefilter_allreduce, the library's lcty_prefilter_allreduce), or of host arrays through the process group.
  * the (genotype, attempt) chains of a solver stage are independent (solvers/solve.rs:1052-1062 deals them to threads):
    contiguous blocks of the stage's genotype list per rank, all-gather of the per-chain likelihoods — on the devices
    through the library's lcty_solve_stage_sharded (Comm.solve_stage), or of host arrays through the process group.
  * a stage of a locus whose reads are sharded needs the possible locations of every read: the location-table rows of the stage's
    alleles are all-gathered between the devices (Comm.solve_stage_read_sharded, lcty_solve_stage_read_sharded); the host-array
    form of that exchange is allgather_read_shards.
torch.distributed is plumbing only (rendezvous, barrier, small host-staged reductions); it is
imported lazily and only when WORLD_SIZE > 1, after liblocityper_hip.so has been loaded.
"""
import os

import numpy as np

_pg = None


def env():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend="gloo"):
    """Initialises the process group when WORLD_SIZE > 1. Returns (rank, local_rank, world)."""
    global _pg
    rank, local_rank, world = env()
    if world > 1 and _pg is None:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
        _pg = dist
    return rank, local_rank, world


def finalize():
    global _pg
    if _pg is not None:
        _pg.destroy_process_group()
        _pg = None


def loci_for_rank(n_loci, rank, world):
    """Round-robin locus assignment: locus l -> rank l % world."""
    return list(range(rank, n_loci, world))


def read_shard(n_pairs, rank, world):
    """Contiguous read shard [lo, hi) of rank `rank`."""
    per = (n_pairs + world - 1) // world
    lo = min(rank * per, n_pairs)
    return lo, min(lo + per, n_pairs)


def barrier():
    if _pg is not None:
        _pg.barrier()


def max_over_ranks(x):
    if _pg is None:
        return float(x)
    import torch
    t = torch.tensor([float(x)], dtype=torch.float64)
    _pg.all_reduce(t, op=_pg.ReduceOp.MAX)
    return float(t[0])


def allreduce_sum_f64(arr):
    """SUM all-reduce of a small f64 vector (partial run_filter scores of read shards)."""
    arr = np.ascontiguousarray(arr, dtype=np.float64)
    if _pg is None:
        return arr
    import torch
    t = torch.from_numpy(arr.copy())
    _pg.all_reduce(t, op=_pg.ReduceOp.SUM)
    return t.numpy()


def chain_block(n_genotypes, rank, world):
    """Block [lo, hi) of a stage's genotype list that rank `rank` solves, and the block size every rank pads to
    (the partition lcty_solve_stage_sharded uses)."""
    per = (n_genotypes + world - 1) // world
    lo = min(rank * per, n_genotypes)
    return lo, min(lo + per, n_genotypes), per


def allgather_chain_liks(local, n_genotypes, attempts):
    """All-gather of the per-chain likelihoods of every rank's block -> [n_genotypes][attempts] on every rank."""
    rank, _, world = env()
    lo, hi, per = chain_block(n_genotypes, rank, world)
    buf = np.full((per, attempts), np.nan)
    buf[:hi - lo] = np.asarray(local, dtype=np.float64).reshape(hi - lo, attempts)
    if _pg is None:
        return buf[:n_genotypes]
    import torch
    mine = torch.from_numpy(buf)
    parts = [torch.empty_like(mine) for _ in range(world)]
    _pg.all_gather(parts, mine)
    return torch.cat(parts).numpy()[:n_genotypes]


def allgather_read_shards(status, weight, unmapped_prob, pa_off, pair_alns):
    """Host-array form of the read-sharded solver exchange (lcty_solve_stage_read_sharded carries the location-table rows of the
    stage's alleles between the devices): every rank's per-pair AllAlignments products all-gathered and concatenated in rank
    order, the pair-alignment offsets moved behind the earlier shards. Returns the arrays of the whole read list."""
    mine = (np.asarray(status), np.asarray(weight), np.asarray(unmapped_prob), np.asarray(pa_off, dtype=np.uint64), np.asarray(pair_alns))
    if _pg is None:
        return mine
    _, _, world = env()
    parts = [None] * world
    _pg.all_gather_object(parts, mine)
    off, base = [np.zeros(1, dtype=np.uint64)], 0
    for q in parts:
        off.append(q[3][1:] + np.uint64(base))
        base += int(q[3][-1])
    return (np.concatenate([q[0] for q in parts]), np.concatenate([q[1] for q in parts]), np.concatenate([q[2] for q in parts]),
            np.concatenate(off), np.concatenate([q[4] for q in parts]))


def make_comm(ctx):
    """RCCL communicator over all ranks (one process per GPU): rank 0 draws the id, the process group (gloo) carries it."""
    from . import api
    rank, _, world = env()
    uid = api.comm_unique_id() if rank == 0 else bytes(api.COMM_ID_BYTES)
    if _pg is not None:
        import torch
        t = torch.tensor(list(uid), dtype=torch.uint8)
        _pg.broadcast(t, src=0)
        uid = bytes(t.tolist())
    return api.Comm(ctx, world, rank, uid)


def gather_objects(obj):
    """List of every rank's `obj` on rank 0 (None elsewhere)."""
    if _pg is None:
        return [obj]
    rank, _, world = env()
    out = [None] * world if rank == 0 else None
    _pg.gather_object(obj, out, dst=0)
    return out
