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
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md)


def pack(seqs):
    off = np.zeros(len(seqs) + 1, np.int64)
    np.cumsum([len(s) for s in seqs], out=off[1:])
    buf = np.concatenate([np.frombuffer(s, np.uint8) if isinstance(s, bytes) else s for s in seqs] +
                         [np.zeros(64, np.uint8)])
    return buf, off


def _gen_span(a):
    from npore_amd import synth
    seed, cnt, ref_len, mixed, first, stride = a
    return synth.make_batch(seed, cnt, ref_len=ref_len, mixed=mixed, first=first, stride=stride)


def make_reads(synth, args, count, rank, world):
    """Reads rank, rank + world, ... of the generator; spans of them on a forked pool when there are many."""
    if count < 2000:
        return synth.make_batch(args.base_seed, count, ref_len=args.ref_len, mixed=args.mixed, first=rank, stride=world)
    import multiprocessing as mp
    nproc = max(1, min(16, (os.cpu_count() or 1) // max(1, world)))
    span = 250
    jobs = [(args.base_seed, min(span, count - k), args.ref_len, args.mixed, rank + k * world, world)
            for k in range(0, count, span)]
    refs, seqs, cigs = [], [], []
    with mp.get_context("fork").Pool(nproc) as pool:
        for r_, s_, c_ in pool.imap(_gen_span, jobs):
            refs += r_; seqs += s_; cigs += c_
    return refs, seqs, cigs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=1000, help="reads per GPU per step")
    ap.add_argument("--ref-len", type=int, default=10_000)
    ap.add_argument("--r", type=int, default=100)
    ap.add_argument("--max-b-rows", type=int, default=20000)
    ap.add_argument("--base-seed", type=int, default=2)
    ap.add_argument("--mixed", action="store_true",
                    help="p_np drawn per read from {0, 0.02, 0.05, 0.15} (SURVEY 8d configs C3 / C4)")
    ap.add_argument("--unique", type=int, default=0,
                    help="generate only this many distinct reads and repeat them to --reads (0 = all distinct)")
    ap.add_argument("--cpu-sample", type=int, default=64, help="reads timed on one host core with the oracle (~10 s)")
    ap.add_argument("--cpu-threads", type=int, default=32,
                    help="worker processes for the whole-batch CPU run (each holds a 241 MB state matrix at r=100; 0/1 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--tb-kernel", type=int, default=0,
                    help="traceback kernel: 0 = chosen by batch size, 1 = windowed, 2 = row per hop (experiments)")
    ap.add_argument("--inflight", type=int, default=1,
                    help="batches in flight per GPU: each has its own context (stream + work buffers) and host thread, "
                         "so one batch's traceback / gather and the next one's preparation run beside a fill kernel")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    # ---- synthetic batch for this rank (reads rank, rank+world, ... : round-robin by index), generated
    # before anything touches the GPU so that large batches can use a forked worker pool
    from npore_amd import synth
    n = args.reads
    n_uniq = min(args.unique, n) if args.unique > 0 else n
    refs, seqs, cigs = make_reads(synth, args, n_uniq, rank, world)
    if n_uniq < n:
        rep = [k % n_uniq for k in range(n)]
        refs = [refs[k] for k in rep]; seqs = [seqs[k] for k in rep]; cigs = [cigs[k] for k in rep]

    import torch
    from npore_amd import _lib, aln
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible (the HIP path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ    # launched by torch.distributed.run
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)                 # "nccl" is RCCL on ROCm

    sub, nps, _, _ = aln.load_default_tables()
    n_ctx = max(1, args.inflight)
    ctxs = [aln.Context(sub, nps, max_n=6, max_l=100, device=local) for _ in range(n_ctx)]
    ctx = ctxs[0]
    if args.tb_kernel:
        for c in ctxs:
            c.set("traceback_kernel", args.tb_kernel)
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

    def step(j=0):
        o, ln, st_ = outs[j]
        rc = lib.npore_align_batch_device(
            ctxs[j].handle, n, d_rb.data_ptr(), d_ro.data_ptr(), d_sb.data_ptr(), d_so.data_ptr(),
            d_cb.data_ptr(), d_co.data_ptr(), 5.0, 1.0, args.max_b_rows, args.r,
            o.data_ptr(), d_oo.data_ptr(), ln.data_ptr(), st_.data_ptr(), None, 1)
        if rc != 0:
            raise RuntimeError(f"npore_align_batch_device: {rc} {_lib.last_error()}")

    def run_steps(k):
        """Exactly k steps; with several batches in flight, host thread j drives steps j, j + n_ctx, ... on its
        own context (the library call blocks its thread and releases the GIL)."""
        times = [[] for _ in range(n_ctx)]

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
                                  {"elapsed": elapsed}, device=dev)
    elapsed = maxes["elapsed"]
    n_bad = int(sums["bad"])
    total_reads = int(sums["reads"])
    assert total_reads == n * world * args.steps
    value = total_reads / elapsed

    # ---- roofline of the dominant kernel (fill): algorithmic bytes per launch / measured duration
    out_len = d_len.cpu().numpy()
    W = 2 * args.r + 1
    bytes_alg = sum(4 * (len(s) + len(r_) + 1) * W + 2 * (len(s) + len(r_)) + int(ol)
                    for s, r_, ol in zip(seqs, refs, out_len))
    if n_uniq < n:
        # repeated reads: every copy must give the string of its original (checked for all lengths, 256 strings)
        assert np.array_equal(out_len, out_len[np.arange(n) % n_uniq]), "copies of one read differ in length"
        for k in np.random.default_rng(0).integers(n_uniq, n, 256):
            a, b = int(oo[k]), int(oo[k % n_uniq])
            assert torch.equal(d_out[a:a + int(out_len[k])], d_out[b:b + int(out_len[k])]), "copies of one read differ"
    fill_avg_ms = float(np.mean(fill_ms))
    achieved = bytes_alg / (fill_avg_ms * 1e-3) / 1e9
    # HBM traffic per launch from the committed rocprofv3 PMC passes of this same default command
    # (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE, KB -> bytes); null for other workloads
    traffic = None
    prof = os.path.join(REPO, "profiles", "r01_fill_pmc_summary.json")
    if (n, args.ref_len, args.r, args.max_b_rows, args.base_seed) == (1000, 10_000, 100, 20000, 2) and os.path.exists(prof):
        pm = json.load(open(prof))
        traffic = int((2 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024)
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "kernel": "fill_kernel", "kernel_ms": round(fill_avg_ms, 3),
                "bytes_alg_per_launch": int(bytes_alg)}

    # ---- CPU baseline: the oracle (plain-C port of the reference DP), bounded sample of the same batch:
    # one host core on the first --cpu-sample reads, then --cpu-threads cores on the whole batch (every read is
    # independent: the reference's own parallelism is a process pool over reads); every string is compared
    # with the GPU output
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:      # reported at N=1 only
        import oracle
        oracle.build()
        out_host = d_out.cpu().numpy()
        n_chk = min(n, max(args.cpu_sample, 10_000_000 // max(1, args.ref_len)))
        got_all = [out_host[oo[i]:oo[i] + out_len[i]].tobytes().decode() for i in range(n_chk)]
        k = min(args.cpu_sample, n)
        tc = time.perf_counter()
        want, st = oracle.align_batch(refs[:k], seqs[:k], cigs[:k], sub, nps, max_b_rows=args.max_b_rows, r=args.r)
        dt1 = time.perf_counter() - tc
        if got_all[:k] != want:
            raise RuntimeError("bench.py: GPU output differs from the oracle on the CPU sample")
        cpu = {"value": round(k / dt1, 3), "unit": "reads/s", "cores": 1, "kind": "port",
               "sample": f"first {k} reads of the same batch, oracle/npore_oracle.c single thread, "
                         f"checked equal to the GPU output; host has {os.cpu_count()} cores"}
        nt = min(args.cpu_threads, os.cpu_count() or 1)
        if nt > 1:
            tc = time.perf_counter()
            # bounded: about 10 M bases of reads (1 000 reads of 10 kb) keep this leg within ~15 s
            m = max(1, min(n, 10_000_000 // max(1, args.ref_len)))
            want, st = oracle.align_batch_procs(refs[:m], seqs[:m], cigs[:m], sub, nps, nt,
                                                max_b_rows=args.max_b_rows, r=args.r)
            dtn = time.perf_counter() - tc
            if got_all[:m] != want:
                raise RuntimeError("bench.py: GPU output differs from the oracle on the whole batch")
            cpu["all_reads"] = {"value": round(m / dtn, 2), "unit": "reads/s", "cores": nt,
                                "sample": f"{'all' if m == n else 'first'} {m} reads of the batch on {nt} worker processes (the reference's own "
                                          f"parallelism: a pool over reads), every string equal to the GPU output"}

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
                       "reads_per_gpu": n, "ref_len": args.ref_len, "r": args.r, "parallelism": f"reads x{world}",
                       "batches_in_flight": n_ctx},
            "roofline": roofline, "cpu_baseline": cpu,
            "stage_ms": {"fill": round(fill_avg_ms, 2), "traceback_gather": round(float(np.mean(tb_ms)), 2),
                         "prep": round(float(np.mean(prep_ms)), 2)},
            "bad_reads": n_bad,
        }
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
