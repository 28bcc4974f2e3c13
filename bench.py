#!/usr/bin/env python3
"""bench.py -- realigned reads/s of the align() hot path on MI355X.

A "step" is one pass of the whole path (path conversion, n-polymer annotation,
banded DP fill, traceback, output gather) over one batch of synthetic ONT-like
reads that is already resident in HBM.  Default workload = BASELINE.json
configs[1]: 1 000 reads of 10 kb, band half-width r=100 ("band=100"),
guppy5_stats penalties, max_b_rows=20000.
With --gpus N every rank runs the same-sized, different batch (weak scaling,
reads dealt by index); the only collective is the final max/sum reduction.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Order of business (one JSON line on rank 0 at the end):
  1. synthetic reads of this rank                                   (host, before anything touches HIP)
  2. CPU baseline: oracle/ (plain-C port of the reference DP) on one core and on the host's cores -- forked
     worker pools, so this also runs BEFORE the GPU runtime is initialised; N=1 only
  3. W warm-up steps, then EXACTLY K timed steps, device-resident    -> value, ms_per_step, roofline
  4. the same batch through the host-buffer entry point (pinned host memory in and out) -> value_pcie_inclusive
  5. a sustained leg: device-resident steps for >= --sustain seconds -> sustained (the GPU is busy long enough
     for an outside sampler to see it; not part of `value`), and `production_default`: one second of the tool's
     default band (r=30) on 4 000 reads, whole path pipelined -- the number a user of realign.py sees per GPU
  6. every CPU string compared with the GPU's.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md)
N_SIMD = 256 * 4        # 256 CUs x 4 SIMDs; a wave64 VALU instruction occupies its SIMD for 4 cycles
PMC_SUMMARY = "r05_fill_pmc_summary.json"   # profiles/: the committed rocprofv3 --pmc summary of the default command (scripts/profile_bench.sh)


def pack(seqs):
    off = np.zeros(len(seqs) + 1, np.int64)
    np.cumsum([len(s) for s in seqs], out=off[1:])
    buf = np.concatenate([np.frombuffer(s, np.uint8) if isinstance(s, bytes) else s for s in seqs] +
                         [np.zeros(64, np.uint8)])
    return buf, off


def make_reads(synth, args, count, rank, world, ref_len=None, mixed=None, base_seed=None):
    """Reads rank, rank + world, ... of the generator; spans of them on a pool of SPAWNED workers when there are many
    (fresh interpreters that only import numpy: under rocprofv3 the profiler's library has initialised the GPU in
    this process before main() runs, and a forked child of such a process must not exist)."""
    ref_len = args.ref_len if ref_len is None else ref_len
    mixed = args.mixed if mixed is None else mixed
    base_seed = args.base_seed if base_seed is None else base_seed
    if count < 2000:
        return synth.make_batch(base_seed, count, ref_len=ref_len, mixed=mixed, first=rank, stride=world)
    import multiprocessing as mp
    nproc = max(1, min(16, host_cpus()["usable"] // max(1, world)))
    span = 250
    jobs = [(base_seed, min(span, count - k), ref_len, mixed, rank + k * world, world)
            for k in range(0, count, span)]
    refs, seqs, cigs = [], [], []
    with mp.get_context("spawn").Pool(nproc) as pool:
        for r_, s_, c_ in pool.imap(synth.make_span, jobs):
            refs += r_; seqs += s_; cigs += c_
    return refs, seqs, cigs


def host_cpus():
    """What this process may use of the host: logical CPUs, its affinity mask, a cgroup CPU quota if any."""
    n = os.cpu_count() or 1
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = n
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    usable = min(n, aff)
    return {"cpu_count": n, "affinity": aff, "cgroup_cpus": quota, "usable": usable}


def host_mem_available():
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    except OSError:
        pass
    for path in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            t = open(path).read().strip()
            if t != "max":
                lim = int(t)
                used = 0
                try:
                    used = int(open(path.replace("memory.max", "memory.current")
                                    .replace("memory.limit_in_bytes", "memory.usage_in_bytes")).read())
                except (OSError, ValueError):
                    pass
                avail = min(avail, lim - used) if avail is not None else lim - used
            break
        except (OSError, ValueError):
            continue
    return avail


def csrc_sha():
    """Digest of the DEVICE sources the library's kernels are built from (what a committed PMC profile is valid for):
    the kernel headers and the generated step assembly -- not the host side of the library (npore_api.cpp, hostio.hpp,
    glue.hpp ...), which changes without touching a kernel."""
    h = hashlib.sha256()
    d = os.path.join(REPO, "npore_amd", "csrc")
    for f in ("annot_wave.hpp", "cell.hpp", "fill_step_asm.inc", "kernels.hpp", "layout.hpp", "prep_kernels.hpp", "std_stream.hpp"):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


DIGESTS = os.path.join(REPO, "tests", "golden", "fullsize_digests.npz")
# (base_seed, mixed, ref_len, r, max_b_rows) -> name in the digest file (tests/golden/make_fullsize_digests.py)
DIGEST_CONFIGS = {(2, False, 10_000, 100, 20000): "c2", (2, False, 10_000, 30, 20000): "r30", (3, True, 10_000, 100, 20000): "c3",
                  (4, True, 10_000, 100, 20000): "c4", (5, False, 50_000, 200, 20000): "c5"}
_DIGESTS = None


def digest_parity(key, read_index, out_host, oo, out_len):
    """Every output string of this rank whose read has a COMMITTED digest of the pinned oracle (tests/golden/
    fullsize_digests.npz: per read index the length and sha256[:16] of the oracle's align() string, made in the build
    container by tests/golden/make_fullsize_digests.py) against that digest -- data, not the oracle: this works on any
    rank of any world size, costs a hash per string and needs nothing of oracle/.  read_index[k] = generator index of
    local read k.  Returns (compared, bad)."""
    global _DIGESTS
    name = DIGEST_CONFIGS.get(key)
    if name is None or not os.path.exists(DIGESTS):
        return 0, 0
    if _DIGESTS is None:
        _DIGESTS = np.load(DIGESTS)
    if name + "_idx" not in _DIGESTS:
        return 0, 0
    idx = _DIGESTS[name + "_idx"].astype(np.int64)
    order = np.argsort(idx)
    idx_s = idx[order]
    ln, dg = _DIGESTS[name + "_len"][order], _DIGESTS[name + "_dig"][order]
    read_index = np.asarray(read_index, np.int64)
    pos = np.searchsorted(idx_s, read_index)
    pos[pos >= len(idx_s)] = 0
    hit = idx_s[pos] == read_index
    compared = bad = 0
    for k in np.nonzero(hit)[0]:
        a, l = int(oo[k]), int(out_len[k])
        ok = l == int(ln[pos[k]]) and int(hashlib.sha256(out_host[a:a + l].tobytes()).hexdigest()[:16], 16) == int(dg[pos[k]])
        compared += 1
        bad += 0 if ok else 1
    return compared, bad


def cpu_baseline(args, refs, seqs, cigs, sub, nps):
    """The oracle (plain-C port of the reference's align(), proven equal to the Cython build in the build
    container) on the GPU box's host cores: one core on a sample, then a process pool over reads -- the
    reference's own parallelism (mp.Pool() over reads, src/realign.py:110-114) -- at several pool sizes up to
    every usable core.  Returns (json dict, {read index: expected string}) for the later comparison."""
    import oracle
    oracle.build()
    n = len(refs)
    want = {}
    k = min(args.cpu_sample, n)
    tc = time.perf_counter()
    got, _ = oracle.align_batch(refs[:k], seqs[:k], cigs[:k], sub, nps, max_b_rows=args.max_b_rows, r=args.r)
    dt1 = time.perf_counter() - tc
    want.update(enumerate(got))
    hc = host_cpus()
    cpu = {"value": round(k / dt1, 3), "unit": "reads/s", "cores": 1, "kind": "port",
           "sample": f"first {k} reads of the same batch, oracle/npore_oracle.c single thread, "
                     f"every string compared with the GPU output",
           "host": {kk: hc[kk] for kk in ("cpu_count", "affinity", "cgroup_cpus")}}
    # k = Cython / port, measured in the build container where the reference can be compiled
    kfile = os.path.join(REPO, "tests", "golden", "k_cython_over_port.json")
    kval = None
    if os.path.exists(kfile):
        kj = json.load(open(kfile))["by_r"]
        rk = min(kj, key=lambda q: abs(int(q) - args.r))
        kval = float(kj[rk]["k"])
        cpu["k_cython_over_port"] = {"k": kval, "measured_at_r": int(rk), "source": "tests/golden/k_cython_over_port.json "
                                     "(tests/golden/measure_k.py, build container: reference Cython vs this port, same reads)"}
        cpu["cython_equivalent"] = {"value": round(k / dt1 * kval, 3), "unit": "reads/s", "cores": 1}
    share = (hc["cgroup_cpus"] or hc["usable"])
    cpu["host_note"] = (f"this process may use {share:g} of the host's {hc['cpu_count']} logical CPUs (cgroup quota of a 1-GPU lease); "
                        f"`all_reads` is the best pool inside that share, NOT a whole host: scaled linearly to all {hc['cpu_count']} CPUs "
                        f"(optimistic for the CPU: the port is memory-bound at many workers) it would be x{hc['cpu_count'] / share:.1f}, "
                        "see `all_reads.whole_host_linear_estimate` and `gpu_speedup.vs_cython_equivalent_whole_host_linear_estimate`")
    if args.cpu_threads != 1:
        state_bytes = 60 * (args.max_b_rows + 1) * (2 * args.r + 1)     # the reference's / the port's state matrix per worker
        avail = host_mem_available()
        cap = hc["usable"] if args.cpu_threads <= 0 else min(args.cpu_threads, hc["usable"])
        if avail:
            cap = max(1, min(cap, int(0.6 * avail // max(1, state_bytes))))
        sizes = sorted({p for p in (16, 32, 64, 128, 256, 512) if p < cap} | {cap})
        if args.cpu_threads > 1:
            sizes = [cap]
        sweep = []
        for p in sizes:
            m = min(n, max(128, 4 * p))
            tc = time.perf_counter()
            got, _ = oracle.align_batch_procs(refs[:m], seqs[:m], cigs[:m], sub, nps, p,
                                              max_b_rows=args.max_b_rows, r=args.r)
            dt = time.perf_counter() - tc
            want.update(enumerate(got))
            sweep.append({"procs": p, "reads": m, "value": round(m / dt, 2)})
            if time.perf_counter() - tc > 40:      # bounded: never more than a couple of slow points
                break
        best = max(sweep, key=lambda e: e["value"])
        cpu["all_reads"] = {"value": best["value"], "unit": "reads/s", "cores": best["procs"],
                            "sample": f"first {best['reads']} reads of the batch on a pool of {best['procs']} worker processes "
                                      f"(the reference's own parallelism; each worker holds a {state_bytes >> 20} MB state matrix); "
                                      f"best of the pool sizes in `sweep`, every string compared with the GPU output",
                            "sweep": sweep}
        scale = hc["cpu_count"] / share
        cpu["all_reads"]["whole_host_linear_estimate"] = round(best["value"] * scale, 1)
        if kval is not None:
            cpu["all_reads"]["cython_equivalent"] = round(best["value"] * kval, 2)
            cpu["all_reads"]["cython_equivalent_whole_host_linear_estimate"] = round(best["value"] * kval * scale, 1)
    return cpu, want


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reads", type=int, default=1000, help="reads per GPU per step")
    ap.add_argument("--ref-len", type=int, default=10_000)
    ap.add_argument("--r", "--band", dest="r", type=int, default=100,
                    help="band half-width (--band: the spelling to use under torch.distributed.run, whose own "
                         "argument parser rejects --r as an ambiguous abbreviation of its --rdzv-* / --role options)")
    ap.add_argument("--max-b-rows", type=int, default=20000)
    ap.add_argument("--base-seed", type=int, default=2)
    ap.add_argument("--mixed", action="store_true",
                    help="p_np drawn per read from {0, 0.02, 0.05, 0.15} (SURVEY 8d configs C3 / C4)")
    ap.add_argument("--unique", type=int, default=0,
                    help="generate only this many distinct reads and repeat them to --reads (0 = all distinct)")
    ap.add_argument("--cpu-sample", type=int, default=64, help="reads timed on one host core with the oracle (~10 s)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="whole-host CPU leg: 0 = sweep pool sizes up to every usable core (memory permitting), "
                         "1 = skip, N > 1 = that many worker processes only")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--pcie-steps", type=int, default=10, help="steps of the host-buffer (PCIe-inclusive) leg; 0 = skip")
    ap.add_argument("--sustain", type=float, default=5.0, help="seconds of the sustained device-resident leg; 0 = skip")
    ap.add_argument("--production", type=float, default=1.0,
                    help="seconds of the `production_default` leg: the tool's default band (r=30, max_b_rows=20000, reference "
                         "src/realign.py:46-51) on one full launch of the fill kernel (4 000 reads of 10 kb), pipelined; 0 = skip")
    ap.add_argument("--production-reads", type=int, default=4000, help="reads per GPU of that leg (tests use fewer)")
    ap.add_argument("--solo-steps", type=int, default=3,
                    help="synchronous steps after the timed region that time the fill kernel on its own (the roofline's duration); "
                         "0 = use the timed region's events (profiler runs)")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="1 (default): steps are enqueued without waiting (sync=0) -- inside the context the next batch is "
                         "prepared and the previous one traced back while the fill kernel works on the current one; the "
                         "timed region ends when the last step has completed.  0: every step waits for its batch")
    ap.add_argument("--coresident", type=int, default=1,
                    help="1 (default): a group of reads that overlaps another one's fill kernel is prepared and gathered "
                         "by kernel shapes that fit beside the fill's workgroups; 0: always the stand-alone ones")
    ap.add_argument("--fill-streams", type=int, default=0, help="1: consecutive fill kernels on one stream, 2: on two (the default; experiments)")
    ap.add_argument("--force-chunks", type=int, default=0, help="chunks per fill workgroup (experiments; 0 = automatic)")
    ap.add_argument("--inflight", type=int, default=1,
                    help="batches in flight per GPU: each has its own context (stream + work buffers) and host thread, "
                         "so one batch's traceback / gather and the next one's preparation run beside a fill kernel")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)

    # ---- 1. synthetic batch for this rank (reads rank, rank+world, ... : round-robin by index)
    from npore_amd import synth, aln
    n = args.reads
    n_uniq = min(args.unique, n) if args.unique > 0 else n
    t_setup0 = time.perf_counter()
    refs, seqs, cigs = make_reads(synth, args, n_uniq, rank, world)
    if n_uniq < n:
        rep = [k % n_uniq for k in range(n)]
        refs = [refs[k] for k in rep]; seqs = [seqs[k] for k in rep]; cigs = [cigs[k] for k in rep]
    sub, nps, _, _ = aln.load_default_tables()
    prod_reads = None
    if args.production > 0:
        prod_reads = make_reads(synth, args, args.production_reads, rank, world, ref_len=10_000, mixed=False, base_seed=2)
    t_generate = time.perf_counter() - t_setup0          # host-side input generation of this rank: BEFORE the timed region

    # ---- 2. CPU baseline (forked pools: before the HIP runtime exists in this process)
    cpu, cpu_want = None, {}
    if rank == 0 and world == 1 and not args.no_cpu:      # reported at N=1 only
        cpu, cpu_want = cpu_baseline(args, refs, seqs, cigs, sub, nps)

    import torch
    from npore_amd import _lib
    n_dev = torch.cuda.device_count()
    if n_dev == 0 or not torch.cuda.is_available():
        print("bench.py: no GPU visible (the HIP path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    # one rank per GPU; on a box with fewer GPUs than ranks (rehearsing the N > 1 path on one card) ranks share
    # devices, and the counter reduction then goes over gloo: RCCL needs one device per rank
    dev_index = local % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ    # launched by torch.distributed.run
    backend = None
    if use_dist:
        import torch.distributed as dist
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        backend = "nccl" if local_world <= n_dev else "gloo"           # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    n_ctx = max(1, args.inflight)
    ctxs = [aln.Context(sub, nps, max_n=6, max_l=100, device=dev_index) for _ in range(n_ctx)]
    ctx = ctxs[0]
    for c in ctxs:
        c.set("coresident", args.coresident)
        if args.fill_streams:
            c.set("fill_streams", args.fill_streams)
        if args.force_chunks:
            c.set("force_chunks", args.force_chunks)
    lib = _lib.load()

    rb, ro = pack(refs)
    sb, so = pack(seqs)
    cb, co = pack(cigs)
    oo = np.zeros(n + 1, np.int64)
    np.cumsum([len(a) + len(b) for a, b in zip(refs, seqs)], out=oo[1:])
    t = lambda a: torch.from_numpy(a).to(dev)
    d_rb, d_ro, d_sb, d_so, d_cb, d_co, d_oo = map(t, (rb, ro, sb, so, cb, co, oo))
    # one set of outputs per batch in flight (inputs are read-only and shared)
    outs = [(torch.zeros(int(oo[-1]) + 64, dtype=torch.uint8, device=dev), torch.zeros(n, dtype=torch.int64, device=dev),
             torch.zeros(n, dtype=torch.int32, device=dev)) for _ in range(n_ctx)]
    d_out, d_len, d_st = outs[0]

    pipelined = bool(args.pipeline) and n_ctx == 1

    def step(j=0, sync=None):
        o, ln, st_ = outs[j]
        rc = lib.npore_align_batch_device(
            ctxs[j].handle, n, d_rb.data_ptr(), d_ro.data_ptr(), d_sb.data_ptr(), d_so.data_ptr(),
            d_cb.data_ptr(), d_co.data_ptr(), 5.0, 1.0, args.max_b_rows, args.r,
            o.data_ptr(), d_oo.data_ptr(), ln.data_ptr(), st_.data_ptr(), None,
            (0 if pipelined else 1) if sync is None else sync)
        if rc != 0:
            raise RuntimeError(f"npore_align_batch_device: {rc} {_lib.last_error()}")

    def run_steps(k):
        """Exactly k steps, all complete on return; the stage times (HIP events on the library's streams) of the
        groups of reads they were made of.  Pipelined: the steps are enqueued back to back on one context.  With
        several contexts (--inflight), host thread j drives steps j, j + n_ctx, ... on its own context (the library
        call blocks its thread and releases the GIL)."""
        if k == 0:
            return []
        times = [[] for _ in range(n_ctx)]
        if pipelined:
            t0_ = ctx.total_timing()
            for _ in range(k):
                step(0)
            ctx.wait()
            t1_ = ctx.total_timing()
            g = max(1.0, t1_["launches"] - t0_["launches"])
            per = tuple((t1_[q] - t0_[q]) / g * (g / k) for q in ("fill_ms", "traceback_ms", "dev_prep_ms"))
            return [per] * k

        def worker(j):
            for _ in range(j, k, n_ctx):
                step(j)
                tm = ctxs[j].timing()       # HIP events recorded on the library's own stream
                times[j].append((tm["fill_ms"], tm["traceback_ms"], tm["dev_prep_ms"]))
        if n_ctx == 1:
            worker(0)
        else:
            import threading
            errs = []

            def guarded(j):
                try:
                    worker(j)
                except BaseException as e:      # noqa: BLE001 -- re-raised on the main thread
                    errs.append(e)
            th = [threading.Thread(target=guarded, args=(j,)) for j in range(n_ctx)]
            for x in th:
                x.start()
            for x in th:
                x.join()
            if errs:
                raise errs[0]
        return [t_ for per in times for t_ in per]

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- 3. the timed region: W warm-up steps, then exactly K steps between barriers
    torch.cuda.synchronize()     # the library's streams are not ordered with torch's: inputs / zero-fills are complete
    warmup = max(args.warmup, n_ctx) if args.warmup else 0        # every context warmed up (buffers allocated)
    run_steps(warmup)
    barrier()
    t0 = time.perf_counter()
    stage = run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    assert len(stage) == args.steps
    for o in outs[1:min(n_ctx, args.steps + warmup)]:             # every batch in flight produced the same strings
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1])
    fill_ms, tb_ms, prep_ms = ([x[i] for x in stage] for i in range(3))
    # the only collective of the job: sum of counters, max of the elapsed time
    from npore_amd.dist import reduce_counters
    sums, maxes = reduce_counters({"reads": n * args.steps, "bad": int(sum((o[2] != 0).sum().item() for o in outs))},
                                  {"elapsed": elapsed, "generate": t_generate, "until_timed": t0 - t_setup0},
                                  device=dev if backend == "nccl" else None)
    elapsed = maxes["elapsed"]
    assert t_setup0 + t_generate <= t0                   # (generation is over before the first barrier of the timed region)
    n_bad = int(sums["bad"])
    total_reads = int(sums["reads"])
    assert total_reads == n * world * args.steps
    value = total_reads / elapsed
    out_len = d_len.cpu().numpy()
    out_host = d_out.cpu().numpy()
    # ---- 3a. EVERY rank proves its strings: each output whose (seed, index, r, max_b_rows) has a committed digest of
    # the pinned oracle is compared with it (any world size; the per-rank counts are summed below)
    par_n, par_bad = digest_parity((args.base_seed, bool(args.mixed), args.ref_len, args.r, args.max_b_rows),
                                   [rank + (k % n_uniq) * world for k in range(n)], out_host, oo, out_len)
    parity = {"main": {"strings_compared": par_n, "strings_bad": par_bad}}

    # ---- 3b. the dominant kernel on its own: three more steps, each complete before the next is enqueued.  In the
    # timed region consecutive fill launches OVERLAP (the next one's workgroups move onto the CUs the previous one's
    # leave), so the HIP events around a launch there also span its wait for room; these launches have the GPU to
    # themselves, as under rocprofv3 (which serialises dispatches) -- the duration the roofline is priced with
    # A call that the library cuts into several groups of reads (C3: 100 000 reads) overlaps the fill launches of
    # consecutive groups itself, and the stage clock sums their event-to-event times: for these steps the fill launches
    # go to ONE stream, so that the sum is a sum of launches that do not share the GPU with another fill (the light
    # kernels of the neighbouring groups still run beside them: the figure is an upper bound of "alone")
    fill_solo = []
    solo_groups = 1
    for q in range(args.solo_steps):
        step(0, sync=1)
        t_ = ctx.timing()
        solo_groups = max(1, int(round(t_["launches"])))
        if solo_groups > 1 and not args.fill_streams:
            if q == 0:
                ctx.set("fill_streams", 1)
                step(0, sync=1)
                t_ = ctx.timing()
        fill_solo.append(t_["fill_ms"])
    if solo_groups > 1 and not args.fill_streams:
        ctx.set("fill_streams", 2)

    # ---- 4. PCIe-inclusive: the same batch through the host-buffer entry point, pinned host memory both ways
    pcie = None
    if args.pcie_steps > 0:
        pin = lambda a: torch.from_numpy(a).pin_memory()
        p_rb, p_sb, p_cb = pin(rb), pin(sb), pin(cb)
        p_out = torch.zeros(int(oo[-1]) + 64, dtype=torch.uint8).pin_memory()
        p_len = torch.zeros(n, dtype=torch.int64).pin_memory()
        p_st = torch.zeros(n, dtype=torch.int32).pin_memory()

        host_fn = lib.npore_align_batch_async if pipelined else lib.npore_align_batch

        def host_step():
            rc = host_fn(ctx.handle, n, p_rb.data_ptr(), ro.ctypes.data, p_sb.data_ptr(), so.ctypes.data,
                         p_cb.data_ptr(), co.ctypes.data, 5.0, 1.0, args.max_b_rows, args.r,
                         p_out.data_ptr(), oo.ctypes.data, p_len.data_ptr(), p_st.data_ptr())
            if rc != 0:
                raise RuntimeError(f"npore_align_batch: {rc} {_lib.last_error()}")
        host_step(); host_step(); host_step()         # warm-up: the staging buffers of every work set allocated
        ctx.wait()
        barrier()
        tt0 = ctx.total_timing()
        tp = time.perf_counter()
        for _ in range(args.pcie_steps):              # pipelined: enqueued back to back, uploads / downloads of one
            host_step()                               # batch beside the kernels of its neighbours
        ctx.wait()
        barrier()
        dtp = time.perf_counter() - tp
        tt1 = ctx.total_timing()
        tms = [{k: (tt1[k] - tt0[k]) / args.pcie_steps for k in ("h2d_ms", "d2h_ms")}]
        _, mx = reduce_counters({}, {"e": dtp}, device=dev if backend == "nccl" else None)
        dtp = mx["e"]
        assert np.array_equal(p_len.numpy(), out_len) and np.array_equal(p_out.numpy()[:int(oo[-1])], out_host[:int(oo[-1])]), \
            "host-buffer entry point and device-resident entry point disagree"
        pcie = {"value": round(n * world * args.pcie_steps / dtp, 1), "unit": "reads/s", "steps": args.pcie_steps,
                "ms_per_step": round(dtp / args.pcie_steps * 1e3, 2),
                "h2d_ms": round(float(np.mean([x["h2d_ms"] for x in tms])), 3),
                "d2h_ms": round(float(np.mean([x["d2h_ms"] for x in tms])), 3),
                "h2d_bytes": int(ro[-1] + so[-1] + co[-1] + 4 * 8 * (n + 1)), "d2h_bytes": int(oo[-1] + 12 * n),
                "note": ("npore_align_batch_async" if pipelined else "npore_align_batch") +
                        " on page-locked host buffers: H2D of bases + CIGARs, the whole path, D2H of the strings"}

    # ---- 5. sustained leg
    sustained = None
    if args.sustain > 0:
        per = max(1, args.steps)
        barrier()
        ts = time.perf_counter()
        done = 0
        while True:
            run_steps(per)
            done += per
            torch.cuda.synchronize()
            go = torch.tensor([1.0 if time.perf_counter() - ts < args.sustain else 0.0], dtype=torch.float64)
            if use_dist:      # every rank stops after the same number of steps
                go = go.to(dev) if backend == "nccl" else go
                dist.all_reduce(go, op=dist.ReduceOp.MAX)
            if float(go.item()) == 0.0:
                break
        barrier()
        dts = time.perf_counter() - ts
        _, mx = reduce_counters({}, {"e": dts}, device=dev if backend == "nccl" else None)
        sustained = {"seconds": round(mx["e"], 2), "steps": done, "value": round(n * world * done / mx["e"], 1), "unit": "reads/s"}

    # ---- 5b. the production default: r=30, one full launch of single-wave chunks, whole path pipelined
    production = None
    if prod_reads is not None:
        p_refs, p_seqs, p_cigs = prod_reads
        pn = len(p_refs)
        prb, pro = pack(p_refs); psb, pso = pack(p_seqs); pcb, pco = pack(p_cigs)
        poo = np.zeros(pn + 1, np.int64)
        np.cumsum([len(a) + len(b) for a, b in zip(p_refs, p_seqs)], out=poo[1:])
        dp = list(map(t, (prb, pro, psb, pso, pcb, pco, poo)))
        p_out = torch.zeros(int(poo[-1]) + 64, dtype=torch.uint8, device=dev)
        p_len = torch.zeros(pn, dtype=torch.int64, device=dev)
        p_st = torch.zeros(pn, dtype=torch.int32, device=dev)

        def prod_steps(k):
            for _ in range(k):
                rc = lib.npore_align_batch_device(ctx.handle, pn, dp[0].data_ptr(), dp[1].data_ptr(), dp[2].data_ptr(),
                                                  dp[3].data_ptr(), dp[4].data_ptr(), dp[5].data_ptr(), 5.0, 1.0, 20000, 30,
                                                  p_out.data_ptr(), dp[6].data_ptr(), p_len.data_ptr(), p_st.data_ptr(), None, 0)
                if rc != 0:
                    raise RuntimeError(f"npore_align_batch_device: {rc} {_lib.last_error()}")
            ctx.wait()
        torch.cuda.synchronize()
        prod_steps(3)
        barrier()
        tt0 = ctx.total_timing()
        tp0 = time.perf_counter()
        done = 0
        while True:
            prod_steps(16)
            done += 16
            go = torch.tensor([1.0 if time.perf_counter() - tp0 < args.production else 0.0], dtype=torch.float64)
            if use_dist:
                go = go.to(dev) if backend == "nccl" else go
                dist.all_reduce(go, op=dist.ReduceOp.MAX)
            if float(go.item()) == 0.0:
                break
        barrier()
        dtp_ = time.perf_counter() - tp0
        tt1 = ctx.total_timing()
        _, mx = reduce_counters({}, {"e": dtp_}, device=dev if backend == "nccl" else None)
        g_ = max(1.0, tt1["launches"] - tt0["launches"])
        p_fill = (tt1["fill_ms"] - tt0["fill_ms"]) / done
        # the fill kernel of this shape with the GPU to itself (synchronous calls): what the step is compared with --
        # the event-to-event times of the pipelined launches above overlap each other and also span their wait for room
        p_solo = []
        for _ in range(max(1, args.solo_steps)):
            rc = lib.npore_align_batch_device(ctx.handle, pn, dp[0].data_ptr(), dp[1].data_ptr(), dp[2].data_ptr(),
                                              dp[3].data_ptr(), dp[4].data_ptr(), dp[5].data_ptr(), 5.0, 1.0, 20000, 30,
                                              p_out.data_ptr(), dp[6].data_ptr(), p_len.data_ptr(), p_st.data_ptr(), None, 1)
            if rc != 0:
                raise RuntimeError(f"npore_align_batch_device: {rc} {_lib.last_error()}")
            p_solo.append(ctx.timing())
        p_fill_alone = float(np.mean([x["fill_ms"] for x in p_solo]))
        p_bytes = sum(4 * (len(s_) + len(r_) + 1) * 61 + 2 * (len(s_) + len(r_)) for s_, r_ in zip(p_seqs, p_refs)) + int(p_len.sum().item())
        assert int((p_st != 0).sum().item()) == 0
        pp_n, pp_bad = digest_parity((2, False, 10_000, 30, 20000), [rank + k * world for k in range(pn)],
                                     p_out.cpu().numpy(), poo, p_len.cpu().numpy())
        parity["production_default"] = {"strings_compared": pp_n, "strings_bad": pp_bad}
        production = {"workload": f"{pn} synthetic 10 kb reads per GPU (base_seed=2), r=30, max_b_rows=20000: the tool's defaults "
                                  "(reference src/realign.py:46-51) at one full launch of the fill kernel; device-resident, pipelined",
                      "value": round(pn * world * done / mx["e"], 1), "unit": "reads/s", "steps": done,
                      "ms_per_step": round(mx["e"] / done * 1e3, 2),
                      "stage_ms_alone": {"fill": round(p_fill_alone, 2),
                                         "traceback_gather": round(float(np.mean([x["traceback_ms"] for x in p_solo])), 2),
                                         "prep": round(float(np.mean([x["dev_prep_ms"] for x in p_solo])), 2),
                                         "note": "synchronous calls behind the leg: every stage has the GPU to itself"},
                      "stage_ms_event_to_event_overlapped": {
                          "fill": round(p_fill, 2), "traceback_gather": round((tt1["traceback_ms"] - tt0["traceback_ms"]) / done, 2),
                          "prep": round((tt1["dev_prep_ms"] - tt0["dev_prep_ms"]) / done, 2),
                          "note": "HIP events around each stage inside the pipelined leg: stages of neighbouring steps run "
                                  "beside each other and wait for each other, so these sum to more than a step"},
                      "exposed_non_fill_ms": round(mx["e"] / done * 1e3 - p_fill_alone, 2),
                      "exposed_non_fill_note": "ms_per_step - fill alone; below zero when the tails of consecutive fill launches overlap",
                      "roofline_frac_hbm": round(p_bytes / (p_fill_alone * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "groups_per_step": round(g_ / done, 2)}
        del dp, p_out

    # ---- roofline of the dominant kernel (fill): algorithmic bytes per launch / measured duration
    W = 2 * args.r + 1
    bytes_alg = sum(4 * (len(s) + len(r_) + 1) * W + 2 * (len(s) + len(r_)) + int(ol)
                    for s, r_, ol in zip(seqs, refs, out_len))
    if n_uniq < n:
        # repeated reads: every copy must give the string of its original (checked for all lengths, 256 strings)
        assert np.array_equal(out_len, out_len[np.arange(n) % n_uniq]), "copies of one read differ in length"
        for k in np.random.default_rng(0).integers(n_uniq, n, 256):
            a, b = int(oo[k]), int(oo[k % n_uniq])
            assert np.array_equal(out_host[a:a + int(out_len[k])], out_host[b:b + int(out_len[k])]), "copies of one read differ"
    fill_region_ms = float(np.mean(fill_ms))       # event to event in the timed region (overlapping launches)
    fill_avg_ms = float(np.mean(fill_solo)) if fill_solo else fill_region_ms      # (--solo-steps 0: profiler runs, which serialise anyway)
    achieved = bytes_alg / (fill_avg_ms * 1e-3) / 1e9
    shape = ctx.fill_shape(args.r)
    rows_total = sum(len(s) + len(r_) + 1 for s, r_ in zip(seqs, refs))
    try:
        clock_ghz = torch.cuda.get_device_properties(dev).clock_rate / 1e6
    except AttributeError:
        clock_ghz = 2.4
    # the kernel's own ceiling: a chunk is a chain of dependent anti-diagonals; `resident_chunks` of them advance
    # side by side, so kernel time ~ (sum of rows / resident chunks) x time per anti-diagonal
    ns_per_row = fill_avg_ms * 1e6 * min(shape["resident_chunks"], max(1, n)) / max(1, rows_total)
    practical = {"ns_per_antidiagonal": round(ns_per_row, 1), "cycles_per_antidiagonal": round(ns_per_row * clock_ghz, 0),
                 "waves_per_chunk": shape["waves_per_chunk"], "resident_waves_per_cu": shape["resident_waves_per_cu"],
                 "resident_chunks": shape["resident_chunks"], "lds_bytes_per_workgroup": shape["lds_bytes"],
                 "clock_ghz": round(clock_ghz, 3), "rows_total": int(rows_total)}
    # PMC-derived figures (HBM traffic, VALU instructions per launch) cannot be collected from inside this
    # process; they come from the committed rocprofv3 --pmc summary of this same default command, and only while
    # that summary was taken from the kernel sources this library was built from (csrc digest); else null
    traffic, valu, pmc_src = None, None, None
    prof = os.path.join(REPO, "profiles", PMC_SUMMARY)
    if os.path.exists(prof):
        pm = json.load(open(prof))
        same_cmd = (n, args.ref_len, args.r, args.max_b_rows, args.base_seed, args.mixed) == \
                   (pm.get("reads"), pm.get("ref_len"), pm.get("r"), pm.get("max_b_rows"), pm.get("base_seed"), bool(pm.get("mixed")))
        if same_cmd and pm.get("csrc_sha") == csrc_sha():
            traffic = int((2 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024)      # gfx950: FETCH_SIZE counts 64 B as 32
            insts = float(pm["SQ_INSTS_VALU"])
            # a SIMD-32 of gfx950 issues one wave64 VOP2 instruction per 2 cycles once two or more of its waves are ready
            # (MI355X_MICROARCH.md "Wave scheduling"; scripts/microbench/valu_issue.cpp on this pool: 2.15 cycles per
            # v_add with 2 ... 8 waves per SIMD, 4.3 for back-to-back VOP3 / DPP / SDWA / VOPC, 4.3 - 5 for ONE wave alone);
            # rounds 1 - 4 priced this with 4 cycles per instruction, which is what one wave alone gets
            peak = N_SIMD * clock_ghz / 2.0                                       # G wave-instructions / s
            ach = insts / (fill_avg_ms * 1e-3) / 1e9
            wave_steps = float(pm.get("wave_steps") or 0)
            # the clock the chip HELD in the profiled launch (MI355X_MICROARCH.md: GRBM_GUI_ACTIVE / 8 XCDs / kernel
            # time) and the share of the launch's cycles in which a SIMD issued a vector instruction
            # (SQ_ACTIVE_INST_VALU counts quad-cycles, summed over the SIMDs)
            held_ghz = busy = None
            if pm.get("GRBM_GUI_ACTIVE"):
                cyc = float(pm["GRBM_GUI_ACTIVE"]) / 8.0
                held_ghz = cyc / (float(pm["kernel_ms_traced_run"]) * 1e-3) / 1e9
                busy = float(pm["SQ_ACTIVE_INST_VALU"]) * 4.0 / N_SIMD / cyc
            valu = {"bound": "valu_issue", "achieved": round(ach, 1), "peak": round(peak, 1), "unit": "G wave-instr/s",
                    "frac": round(ach / peak, 4), "insts_valu_per_launch": int(insts),
                    "valu_per_wave_step": round(insts / wave_steps, 1) if wave_steps else None,
                    "peak_note": "1 024 SIMD-32 x clock / 2 cycles per wave64 VOP2 instruction (needs >= 2 ready waves per SIMD; "
                                 "VOP3 / DPP / SDWA / compares back to back run at half of it: scripts/microbench/valu_issue.cpp)",
                    "clock_held_ghz": round(held_ghz, 3) if held_ghz else None,
                    "valu_busy_frac_at_held_clock": round(busy, 4) if busy else None,
                    "valu_busy_note": "SQ_ACTIVE_INST_VALU x 4 cycles / SIMD / cycles of the launch: the counter advances once per "
                                      "vector instruction (it equals SQ_INSTS_VALU within 0.5 %), so this prices every instruction "
                                      "at 4 cycles -- an upper bound of the issue pipe's occupancy, not a measurement of it"}
            pmc_src = f"profiles/{PMC_SUMMARY} (csrc {pm['csrc_sha']}): quoted from that committed rocprofv3 --pmc run of this same command, not measured by this process"
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "frac_by_step_throughput": round(bytes_alg / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "kernel": "fill_kernel", "kernel_ms": round(fill_avg_ms, 3),
                "kernel_ms_source": ((f"HIP events around {len(fill_solo)} launches that had the GPU to themselves (after the timed region)"
                                      if solo_groups == 1 else
                                      f"sum of HIP-event times over the {solo_groups} groups of reads of one call ({len(fill_solo)} calls after the timed "
                                      "region), fill launches on one stream: no two fills share the GPU, the neighbouring groups' light kernels run beside them")
                                     if fill_solo else "HIP events around the launches of the timed region"),
                "kernel_ms_in_timed_region": round(fill_region_ms, 3),
                "bytes_alg_per_launch": int(bytes_alg), "valu_issue": valu, "practical": practical, "pmc_source": pmc_src}

    # ---- 5c. the ranks' parity counts, summed (a second tiny reduction, outside every timed region)
    flat = {f"{leg}.{q}": v for leg, d_ in parity.items() for q, v in d_.items()}
    flat, _ = reduce_counters(flat, {}, device=dev if backend == "nccl" else None)
    parity = {leg: {q: int(flat[f"{leg}.{q}"]) for q in d_} for leg, d_ in parity.items()}
    parity["strings_compared"] = sum(d_["strings_compared"] for d_ in parity.values() if isinstance(d_, dict))
    parity["strings_bad"] = sum(d_["strings_bad"] for d_ in parity.values() if isinstance(d_, dict))
    parity["source"] = ("tests/golden/fullsize_digests.npz: per-read (length, sha256[:16]) of the pinned oracle's align() string, "
                        "made in the build container (tests/golden/make_fullsize_digests.py); every rank compares every "
                        "output whose (base_seed, read index, r, max_b_rows) is in the file, counts summed over ranks")

    # ---- 6. every CPU string against the GPU's
    if cpu is not None:
        for i, w in cpu_want.items():
            g = out_host[oo[i]:oo[i] + out_len[i]].tobytes().decode()
            if g != w:
                raise RuntimeError(f"bench.py: GPU output differs from the oracle on read {i}")
        cpu["strings_compared"] = len(cpu_want)
        base1 = cpu["value"]
        sp = {"vs_port_1core": round(value / base1, 1)}
        if "all_reads" in cpu:
            sp["vs_port_all_cores"] = round(value / cpu["all_reads"]["value"], 1)
        if "cython_equivalent" in cpu:
            sp["vs_cython_equivalent_1core"] = round(value / cpu["cython_equivalent"]["value"], 1)
            if "all_reads" in cpu:
                sp["vs_cython_equivalent_all_cores"] = round(value / cpu["all_reads"]["cython_equivalent"], 1)
                sp["vs_cython_equivalent_whole_host_linear_estimate"] = round(
                    value / cpu["all_reads"]["cython_equivalent_whole_host_linear_estimate"], 1)
                sp["note"] = ("`all_cores` = the cores of this lease's cgroup share (cpu_baseline.host_note), not of the host; the "
                              "whole-host figure is a linear extrapolation")
        cpu["gpu_speedup"] = sp

    if rank == 0:
        line = {
            "metric": "realigned reads/sec (10 kb ONT-like)", "value": round(value, 1), "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n} synthetic {args.ref_len // 1000} kb reads per GPU, r={args.r} "
                                   f"(band={args.r}), max_b_rows={args.max_b_rows}, guppy5_stats penalties "
                                   f"(SURVEY 8d generator{', mixed p_np' if args.mixed else ''}, base_seed={args.base_seed}"
                                   f"{f', {n_uniq} distinct reads repeated' if n_uniq < n else ''})",
                       "reads_per_gpu": n, "ref_len": args.ref_len, "r": args.r, "max_b_rows": args.max_b_rows,
                       "base_seed": args.base_seed, "mixed": bool(args.mixed), "parallelism": f"reads x{world}",
                       "batches_in_flight": n_ctx, "pipelined": pipelined, "devices_visible": n_dev,
                       "value_excludes": "H2D/D2H: inputs and outputs stay in HBM across the timed region (the bench contract of "
                                         "`value`); the same batch through page-locked host buffers is `value_pcie_inclusive`",
                       "survey_8d_metric": "SURVEY.md 8(d) words the metric with H2D/D2H INCLUDED: that figure is "
                                           "`value_pcie_inclusive.value` (host buffers in, strings out), reported beside `value`",
                       "reduction_backend": {"nccl": "rccl", "gloo": "gloo (ranks share a device)", None: "none"}[backend],
                       "host_setup": {"generate_s_max_over_ranks": round(maxes["generate"], 2),
                                      "start_to_timed_region_s_max_over_ranks": round(maxes["until_timed"], 2),
                                      "note": "every rank generates its own reads before the first barrier: outside the timed region"}},
            "roofline": roofline, "cpu_baseline": cpu,
            "value_pcie_inclusive": pcie, "sustained": sustained, "production_default": production,
            "stage_ms": {"fill_alone": round(fill_avg_ms, 2), "fill_event_to_event_overlapped": round(fill_region_ms, 2),
                         "traceback_gather": round(float(np.mean(tb_ms)), 2), "prep": round(float(np.mean(prep_ms)), 2),
                         "note": "fill_alone: the kernel with the GPU to itself (what the roofline is priced with); the other "
                                 "three are HIP-event times inside the timed region, where consecutive steps' stages overlap"},
            "bad_reads": n_bad, "parity": parity,
        }
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()
    for c in ctxs:
        c.close()
    if parity["strings_bad"]:
        print(f"bench.py: {parity['strings_bad']} of {parity['strings_compared']} strings differ from the pinned oracle's digests",
              file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
