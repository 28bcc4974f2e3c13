"""Multi-GPU sharding of the read stream: one process per GPU, reads dealt
round-robin by index, no data-path collective (reads are independent); the only
exchange is one small sum/max reduction of counters (RCCL on GPUs, gloo in the
CPU tests).  Counterpart of the reference's process-level data parallelism
(multiprocessing.Pool over reads, src/realign.py:110-114)."""
import os


def world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_indices(n_items, rank, world_size):
    """Indices of the reads this rank owns (round-robin by index)."""
    return range(rank, n_items, world_size)


def reduce_counters(sums, maxes, device=None):
    """All-reduce {name: number}: `sums` summed, `maxes` maximised over ranks.
    No-op when torch.distributed is not initialised (single process)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dict(sums), dict(maxes)
    ks, km = sorted(sums), sorted(maxes)
    ts = torch.tensor([float(sums[k]) for k in ks], dtype=torch.float64, device=device)
    tm = torch.tensor([float(maxes[k]) for k in km], dtype=torch.float64, device=device)
    dist.all_reduce(ts, op=dist.ReduceOp.SUM)
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    return dict(zip(ks, ts.tolist())), dict(zip(km, tm.tolist()))
