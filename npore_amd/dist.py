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
    if ks:                                   # (no collective on an empty tensor: every rank passes the same keys)
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
    if km:
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    return dict(zip(ks, ts.tolist())), dict(zip(km, tm.tolist()))


def _append_file(out, src):
    """Append the open file `src` at the current position of `out` (opened 'r+b' / 'wb', NOT in append mode:
    copy_file_range refuses an O_APPEND destination with EBADF) inside the kernel -- no trip through user space,
    and a reflink where the file system has them; plain reads and writes where that is not available."""
    size = os.fstat(src.fileno()).st_size
    done = 0
    if hasattr(os, "copy_file_range"):
        out.flush()
        pos = out.tell()
        try:
            while done < size:
                k = os.copy_file_range(src.fileno(), out.fileno(), min(size - done, 1 << 30),
                                       offset_src=done, offset_dst=pos + done)
                if k <= 0:
                    break
                done += k
        except OSError:
            pass
        out.seek(pos + done)
    src.seek(done)
    while True:
        buf = src.read(1 << 24)
        if not buf:
            break
        out.write(buf)


def barrier_file_ranks():
    """(rank, world, barrier) for host-only steps of a multi-process run: barrier() is a gloo barrier, created on
    first use when torch.distributed is not initialised yet; a no-op in a single process."""
    rank, world_size, _ = world()
    if world_size == 1:
        return rank, world_size, (lambda: None)

    def barrier():
        import datetime
        import torch.distributed as dist
        if not dist.is_initialized():
            # host-only steps between barriers may take hours (a genome-scale --recalc_cms on one rank)
            dist.init_process_group("gloo", timeout=datetime.timedelta(hours=float(os.environ.get("NPORE_HOST_STEP_TIMEOUT_H", "48"))))
        dist.barrier()
    return rank, world_size, barrier


def rank0_then_all(work):
    """Run `work()` on rank 0 while the other ranks wait; every rank learns whether it succeeded (rank 0 re-raises its
    own exception, the others raise RuntimeError) -- no rank is left at a barrier that never comes.  Single process:
    just runs it.  Returns work()'s result on rank 0, None elsewhere."""
    rank, world_size, barrier = barrier_file_ranks()
    if world_size == 1:
        return work()
    import torch
    import torch.distributed as dist
    barrier()                                         # (creates the gloo group with its long timeout)
    res, err = None, None
    if rank == 0:
        try:
            res = work()
        except BaseException as e:                    # incl. SystemExit: the flag below must still go out
            err = e
    flag = torch.tensor([0.0 if err is None else 1.0], dtype=torch.float64)
    dist.all_reduce(flag, op=dist.ReduceOp.SUM)
    if err is not None:
        raise err
    if flag.item() != 0.0:
        raise RuntimeError("rank 0 failed in a host-only step (see its output)")
    return res


def cgroup_cpus():
    """CPUs a cgroup quota leaves this process (None: no quota).  cgroup v2 cpu.max, v1 cfs_quota / cfs_period."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()[:2]
            return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        return None if q <= 0 else q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
    except (OSError, ValueError):
        return None


def host_threads_per_rank():
    """Host threads a rank should use for the parallel host stages (BGZF inflate, packing, standardisation, SAM
    text): the CPUs this process may really use -- its affinity mask cut down by a cgroup quota (a 16-CPU lease on a
    256-CPU host reports 256 in the mask) -- divided among the ranks of this node."""
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    quota = cgroup_cpus()
    if quota is not None:
        cores = min(cores, max(1, int(quota + 0.5)))
    return max(1, cores // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))


def gather_parts(final_path, out_prefix, n_local):
    """End of a multi-process realign: wait for every rank's part file, let rank 0 append them to the
    final SAM in rank order and remove them.  Returns the total number of reads (on every rank).
    Uses a gloo process group (CPU): the payload is one barrier and one small sum."""
    import torch
    import torch.distributed as dist
    rank, world_size, _ = world()
    own_group = not dist.is_initialized()
    if own_group:
        dist.init_process_group("gloo")
    t = torch.tensor([float(n_local)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)        # also the barrier: every part is complete
    if rank == 0:
        with open(final_path, "r+b") as out:
            out.seek(0, os.SEEK_END)
            for k in range(world_size):
                part = f"{out_prefix}.part{k}.sam"
                with open(part, "rb") as fh:
                    _append_file(out, fh)
                os.remove(part)
    dist.barrier()
    if own_group:
        dist.destroy_process_group()
    return int(t.item())
