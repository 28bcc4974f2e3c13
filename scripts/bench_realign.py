#!/usr/bin/env python3
"""End-to-end BAM -> SAM throughput of the realign path with a stage breakdown.

A synthetic BAM is made from `--reads` reads of the bench generator laid end to end on one contig -- every read its own
draw, qualities one uniform draw per base over phred 0 ... 93 as the reference's fixture generator draws them
(test/generate_bam.py:63,79) -- on a pool of worker processes, BGZF-compressed on a thread pool (zlib level 1).
`--distinct d --const-qual` makes the file of rounds 3 - 4 instead (d reads repeated, constant qualities: a third of the
bytes per read, and blocks that are all matches).  One JSON line: the stages of the native pipeline (open = BGZF inflate + record
index, select, then per batch record fetch + pack | H2D + kernels + D2H | standardise | SAM text | write, overlapped by
npore_bam_realign_file), the GPU's busy share of the wall time, reads/s; the same for a STREAMED handle
(bounded-memory ingest) with the process's peak resident set; and the pure-Python restatement on a few reads.

    python scripts/bench_realign.py [--reads 48000] [--distinct 4000] [--ref-len 10000] [--r 30] [--batch 2000]
"""
import argparse
import json
import os
import resource
import struct
import sys
import tempfile
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from npore_amd import aln, bam, cfg, synth


def bgzf_write(path, data, level=1, threads=16):
    """`data` as a BGZF file (blocks of 0xFF00 bytes compressed on a thread pool: zlib releases the GIL)."""
    def block(p):
        chunk = bytes(data[p:p + 0xFF00])
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = c.compress(chunk) + c.flush()
        return struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 0xFF, 6, 66, 67, 2, len(comp) + 25) + comp + \
            struct.pack("<II", zlib.crc32(chunk), len(chunk))
    with ThreadPoolExecutor(threads) as tp, open(path, "wb") as fh:
        for piece in tp.map(block, range(0, len(data), 0xFF00), chunksize=64):
            fh.write(piece)
        fh.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))


_SEQ_NIB = np.array([15, 1, 2, 4, 8], np.uint8)         # base codes N A C G T -> BAM 4-bit codes (=ACMGRSVTWYHKDBN)
READS_PER_CONTIG = 100_000                              # (a BAM position is 31 bits: the reads lie end to end on contigs of 1 Gbp)


def _encode_span(job):
    """Worker: reads first ... first + count - 1 of the bench generator as BAM records (bytes) + their stretch of the contig.
    Qualities: one uniform draw per base over phred 0 ... 93, as the reference's own fixture generator does
    (test/generate_bam.py:63,79: chr(randint(33, 127)) per base) -- or the constant 20 of rounds 3 - 4 (`const_qual`)."""
    seed, first, count, ref_len, const_qual, per_ctg = job
    dec = np.frombuffer(b"NACGT", np.uint8)
    recs, contig = [], []
    for k in range(first, first + count):
        rf, sq, cg = synth.make_pair(seed, k, ref_len)
        cg = np.frombuffer(cg, np.uint8)
        edges = np.flatnonzero(np.diff(cg)) + 1
        starts = np.concatenate(([0], edges))
        lens = np.diff(np.concatenate((starts, [len(cg)]))).astype(np.uint32)
        opc = np.zeros(256, np.uint32)
        for ch, code in (("I", 1), ("D", 2), ("=", 7), ("X", 8), ("M", 0)):
            opc[ord(ch)] = code
        cig = ((lens << 4) | opc[cg[starts]]).astype("<u4").tobytes()
        nib = _SEQ_NIB[sq]
        if len(nib) & 1:
            nib = np.concatenate((nib, np.zeros(1, np.uint8)))
        packed = ((nib[0::2] << 4) | nib[1::2]).astype(np.uint8).tobytes()
        if const_qual:
            qual = bytes([20]) * len(sq)
        else:
            qual = np.random.Generator(np.random.PCG64(seed * 7919 + k)).integers(0, 94, len(sq), dtype=np.uint8).tobytes()
        name = f"read{k}".encode() + b"\0"
        body = struct.pack("<iiBBHHHiiii", k // per_ctg, (k % per_ctg) * ref_len, len(name), 60, 4680, len(lens), 0, len(sq), -1, -1, 0) + \
            name + cig + packed + qual + b"HPC" + bytes([k % 3])
        recs.append(struct.pack("<i", len(body)) + body)
        contig.append(dec[rf].tobytes())
        assert len(rf) == ref_len
    return b"".join(recs), b"".join(contig)


def build_inputs(tmp, n, distinct, ref_len, seed, const_qual=False, procs=0):
    """The BAM (BGZF, zlib level 1) and FASTA of `n` reads laid end to end on one contig.  distinct = 0 (default): every
    read is its own draw of the generator, made on a pool of worker processes; distinct = d < n: the first d reads'
    record block repeated at the byte level (rounds 3 - 4: 4 000 distinct reads x 24)."""
    import multiprocessing as mp
    distinct = n if distinct <= 0 else min(distinct, n)
    procs = procs or max(1, min(16, len(os.sched_getaffinity(0))))
    span = 250
    jobs = [(seed, k, min(span, distinct - k), ref_len, const_qual, READS_PER_CONTIG) for k in range(0, distinct, span)]
    recs, contig = [], []
    with mp.get_context("spawn").Pool(procs) as pool:
        for r_, c_ in pool.imap(_encode_span, jobs):
            recs.append(r_); contig.append(c_)
    contig = b"".join(contig)
    if distinct < n and n > READS_PER_CONTIG:
        raise SystemExit("--distinct with more than one contig's worth of reads is not supported")
    n_ctg = (distinct + READS_PER_CONTIG - 1) // READS_PER_CONTIG
    ctg_len = [min(READS_PER_CONTIG, distinct - c * READS_PER_CONTIG) * ref_len for c in range(n_ctg)]
    names = ["ctg"] + [f"ctg{c + 1}" for c in range(1, n_ctg)]
    fa = os.path.join(tmp, "ref.fa")
    with open(fa, "wb") as fh:
        arr_all = np.frombuffer(contig, np.uint8)
        at = 0
        for name, ln in zip(names, ctg_len):
            fh.write(b">" + name.encode() + b"\n")
            arr = arr_all[at:at + ln]
            at += ln
            full = len(arr) // 60 * 60
            lines = np.empty((full // 60, 61), np.uint8)
            lines[:, :60] = arr[:full].reshape(-1, 60)
            lines[:, 60] = 10
            fh.write(lines.tobytes())
            if full < len(arr):
                fh.write(arr[full:].tobytes() + b"\n")
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{nm}\tLN:{ln}\n" for nm, ln in zip(names, ctg_len))
    header = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", n_ctg) + \
        b"".join(struct.pack("<i", len(nm) + 1) + nm.encode() + b"\0" + struct.pack("<i", ln) for nm, ln in zip(names, ctg_len))
    body = b"".join(recs)
    reps, rest = divmod(n, distinct)
    q = 0
    for _ in range(rest):                        # the first `rest` records once more
        bs, = struct.unpack_from("<i", body, q); q += 4 + bs
    bp = os.path.join(tmp, "reads.bam")
    bgzf_write(bp, header + body * reps + body[:q], threads=procs)
    return bp, fa, (ctg_len[0] if n_ctg == 1 else [[nm, ln] for nm, ln in zip(names, ctg_len)])


def regions_of(clen):
    """[(contig, start, stop)] of the generated file (one contig: its length; several: [[name, length], ...])"""
    return [("ctg", 0, clen - 1)] if isinstance(clen, int) else [(nm, 0, ln - 1) for nm, ln in clen]


def run_file(ctx, bp, fa, clen, a, out, stream):
    """open -> select -> npore_bam_realign_file; returns the stage dictionary"""
    t0 = time.perf_counter()
    nb, nf = bam.NativeBam(bp, stream=stream, threads=a.threads), bam.NativeFasta(fa)
    t1 = time.perf_counter()
    idx = nb.select(regions_of(clen))
    t2 = time.perf_counter()
    bam.create_header(out, nb)
    bam.realign_native(ctx, nb, nf, idx, out, r=a.r, batch_reads=a.batch, threads=a.threads)
    t3 = time.perf_counter()
    ft = nb.file_timing()
    wall = ft["wall_ms"] * 1e-3
    res = {"streamed": bool(nb.streamed), "reads": int(len(idx)), "open_inflate_index_s": round(t1 - t0, 3), "select_s": round(t2 - t1, 3),
           "realign_file_s": round(t3 - t2, 3), "total_s": round(t3 - t0, 3),
           "reads_per_s": round(len(idx) / (t3 - t0), 1), "reads_per_s_batches_only": round(len(idx) / (t3 - t2), 1),
           "stage_sums_s": {k[:-3]: round(v * 1e-3, 3) for k, v in ft.items() if k not in ("wall_ms",)},
           "gpu_busy_fraction_of_batches": round((ft["gpu_kernels_ms"] * 1e-3) / max(wall, 1e-9), 3),
           "gpu_idle_fraction_of_batches": round(1.0 - (ft["gpu_kernels_ms"] * 1e-3) / max(wall, 1e-9), 3),
           "sam_bytes": os.path.getsize(out) if out != "/dev/null" else None,
           "inflated_bam_bytes": int(nb._lib.npore_bam_inflated_size(nb.handle))}
    nb.close(); nf.close()
    return res


def warm_up(ctx, bp, fa, clen, a):
    """One untimed pass over the first three batches: a process allocates the context's three work sets (24 GB of traceback
    words each at this batch size) and the slots' page-locked buffers ONCE, and what that costs is the driver's business --
    on this pool hipMalloc of 24 GB takes 0.3 ms, 0.13 s, 0.7 s or 4 s depending on how much of the card the driver has
    scrubbed (scripts/microbench/alloc_time.py, profiles/r05_alloc_microbench.txt).  Without this the first timed leg carries
    it (31 k ... 88 k reads/s for the same code); reported as `cold_start`."""
    t0 = time.perf_counter()
    nb, nf = bam.NativeBam(bp, one_pass=True, threads=a.threads), bam.NativeFasta(fa)
    t1 = time.perf_counter()
    n, _, _ = nb.realign_sequential(ctx, nf, regions_of(clen), "/dev/null", batch_reads=a.batch, max_reads=3 * a.batch, r=a.r, threads=a.threads)
    t2 = time.perf_counter()
    nb.close(); nf.close()
    return {"reads": int(n), "open_s": round(t1 - t0, 3), "realign_s": round(t2 - t1, 3),
            "note": "the first three batches of the file, one pass, text discarded, in a process that has not touched the GPU's memory yet"}


def cgroup_cpu_stat():
    """{nr_periods, nr_throttled, throttled_usec} of this process's cgroup (v2), or {}: whether the quota of a lease -- not the
    number of CPUs -- stopped the process's threads (a burst of more runnable threads than the quota uses up the period's share
    early, and everything waits for the next period)."""
    try:
        out = {}
        for line in open("/sys/fs/cgroup/cpu.stat"):
            k, v = line.split()
            if k in ("nr_periods", "nr_throttled", "throttled_usec"):
                out[k] = int(v)
        return out
    except OSError:
        return {}


def run_one_pass(ctx, bp, fa, clen, a, out):
    """header-only open -> npore_bam_realign_sequential (what `realign` does for one process and whole-contig regions)"""
    t0 = time.perf_counter()
    nb, nf = bam.NativeBam(bp, one_pass=True, threads=a.threads), bam.NativeFasta(fa)
    t1 = time.perf_counter()
    bam.create_header(out, nb)
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    cg0 = cgroup_cpu_stat()
    n, bad, _ = nb.realign_sequential(ctx, nf, regions_of(clen), out, batch_reads=a.batch, r=a.r, threads=a.threads)
    t2 = time.perf_counter()
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    cg1 = cgroup_cpu_stat()
    cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    ft = nb.file_timing()
    res = {"one_pass": True, "reads": int(n), "open_header_s": round(t1 - t0, 3), "realign_s": round(t2 - t1, 3), "total_s": round(t2 - t0, 3),
           "reads_per_s": round(n / (t2 - t0), 1),
           "host_cpu_s": round(cpu_s, 3), "host_cpu_us_per_read": round(cpu_s / max(n, 1) * 1e6, 1),
           "host_cpus_busy": round(cpu_s / max(t2 - t1, 1e-9), 2),
           "cgroup_throttling": {k: cg1[k] - cg0[k] for k in cg1 if k in cg0} or None,
           "host_cpu_note": "user + system time of this process during the realign call (all threads): what the host stages cost per "
                            "read, and how many CPUs they kept busy on average (the lease's cgroup quota is the ceiling)",
           "stage_sums_s": {k[:-3]: round(v * 1e-3, 3) for k, v in ft.items() if k not in ("wall_ms",)},
           "stage_sums_note": "fetch_pack includes the inflation of every block (once); align_call = waiting for a batch's completion event; "
                              "gpu_kernels / pcie are sums of stage times of groups that overlap on the device",
           "sam_bytes": os.path.getsize(out) if out != "/dev/null" else None, "bad_reads": len(bad)}
    nb.close(); nf.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=48000)
    ap.add_argument("--distinct", type=int, default=0,
                    help="0 (default): every read distinct; d > 0: d distinct reads repeated to --reads (rounds 3 - 4 used 4000)")
    ap.add_argument("--const-qual", action="store_true",
                    help="qualities = the constant 20 (rounds 3 - 4) instead of one uniform draw per base over phred 0 ... 93 "
                         "(what the reference's fixture generator draws: test/generate_bam.py:63,79)")
    ap.add_argument("--ref-len", type=int, default=10000)
    ap.add_argument("--r", type=int, default=30)
    ap.add_argument("--batch", type=int, default=2000)
    ap.add_argument("--py-reads", type=int, default=32)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--tmp", default=None, help="directory for the generated files (default: a temporary one)")
    ap.add_argument("--threads", type=int, default=0, help="host threads of the parallel host stages (0 = the library's default)")
    ap.add_argument("--one-pass-only", action="store_true",
                    help="only the ONE-PASS leg, SAM text to /dev/null: bounded memory and the host's inflate rate on a file of tens of GB")
    ap.add_argument("--streamed-only", action="store_true",
                    help="only the STREAMED leg, SAM text to /dev/null: the bounded-memory demonstration on a file of tens of GB")
    ap.add_argument("--cold", action="store_true",
                    help="no untimed warm-up pass: the first timed leg then carries the process's one-time device and page-locked allocations")
    ap.add_argument("--gen-into", default=None, help=argparse.SUPPRESS)       # (internal: make the inputs in this directory and exit)
    a = ap.parse_args()
    if a.gen_into:
        print(json.dumps(build_inputs(a.gen_into, a.reads, a.distinct, a.ref_len, a.seed, a.const_qual)))
        return
    sub, nps, _, _ = aln.load_default_tables()
    with tempfile.TemporaryDirectory(dir=a.tmp) as tmp:
        t = time.perf_counter()
        # (in a child process of its own -- it runs a pool of workers -- : the generator's GBs of Python byte strings must
        # not count towards this process's peak RSS)
        import subprocess
        gen = subprocess.run([sys.executable, os.path.abspath(__file__), "--gen-into", tmp, "--reads", str(a.reads), "--distinct", str(a.distinct),
                              "--ref-len", str(a.ref_len), "--seed", str(a.seed)] + (["--const-qual"] if a.const_qual else []),
                             capture_output=True, text=True)
        if gen.returncode != 0:
            sys.exit("input generation failed:\n" + gen.stderr[-3000:])
        bp, fa, clen = json.loads(gen.stdout.strip().splitlines()[-1])
        t_gen = time.perf_counter() - t
        ctx = aln.Context(sub, nps)
        cfg.args = argparse.Namespace(max_n=6, max_l=100, regions=regions_of(clen), max_reads=0)
        out = os.path.join(tmp, "out.sam")
        # the STREAMED handle first: ru_maxrss is the peak of the whole process so far, and the resident handle holds
        # the inflated file (the context's page-locked staging and the GPU runtime are in both figures)
        cold = None if a.cold else warm_up(ctx, bp, fa, clen, a)
        rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
        if a.one_pass_only:
            one_pass = run_one_pass(ctx, bp, fa, clen, a, "/dev/null")
            one_pass["cold_start"] = cold
            one_pass["peak_rss_mb"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, 1)
            print(json.dumps({"metric": "BAM->SAM realigned reads/sec (one-pass ingest, SAM text discarded)", "value": one_pass["reads_per_s"],
                              "unit": "reads/s", "reads": a.reads, "distinct_reads": a.reads if a.distinct <= 0 else min(a.distinct, a.reads), "qualities": "constant 20" if a.const_qual else "uniform per base over phred 0 ... 93", "r": a.r, "batch": a.batch,
                              "bam_bytes": os.path.getsize(bp), "one_pass": one_pass, "rss_mb_before_timed_runs": round(rss0 / 1024.0, 1),
                              "input_generation_s": round(t_gen, 1)}))
            ctx.close()
            return
        streamed = run_file(ctx, bp, fa, clen, a, "/dev/null" if a.streamed_only else out + ".s", True)
        streamed["peak_rss_mb"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, 1)
        if a.streamed_only:
            print(json.dumps({"metric": "BAM->SAM realigned reads/sec (streamed ingest, SAM text discarded)", "value": streamed["reads_per_s"],
                              "unit": "reads/s", "reads": a.reads, "distinct_reads": a.reads if a.distinct <= 0 else min(a.distinct, a.reads), "qualities": "constant 20" if a.const_qual else "uniform per base over phred 0 ... 93", "r": a.r, "batch": a.batch,
                              "bam_bytes": os.path.getsize(bp), "streamed": streamed, "rss_mb_before_timed_runs": round(rss0 / 1024.0, 1),
                              "input_generation_s": round(t_gen, 1)}))
            ctx.close()
            return
        one_pass = run_one_pass(ctx, bp, fa, clen, a, out + ".o")
        one_pass["peak_rss_mb"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, 1)
        resident = run_file(ctx, bp, fa, clen, a, out, False)
        resident["peak_rss_mb"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, 1)
        same = os.path.getsize(out) == os.path.getsize(out + ".s") and open(out, "rb").read(1 << 24) == open(out + ".s", "rb").read(1 << 24)
        import hashlib
        digest = lambda f: hashlib.sha256(open(f, "rb").read()).hexdigest()
        same_one_pass = digest(out) == digest(out + ".o")
        # how busy the GPU is in the one-pass run: the indexed resident leg's batch phase runs at the device path's own rate
        # (packed batches -> upload, kernels, download; the host stages keep up there), so that rate x the one-pass wall
        dev_rate = resident["reads_per_s_batches_only"]
        one_pass["gpu_busy_fraction_estimate"] = round(min(1.0, (one_pass["reads"] / dev_rate) / max(one_pass["realign_s"], 1e-9)), 3)
        one_pass["gpu_busy_note"] = ("reads / (the device path's rate = the resident leg's batch phase) / this leg's wall; the rest of the time the "
                                     "GPU waits for the host stages (inflate + pack in front of it, standardise + SAM text behind it)")
        for leg in (resident, streamed):
            leg["gpu_stage_sum_fraction_of_batches"] = leg.pop("gpu_busy_fraction_of_batches")
            leg.pop("gpu_idle_fraction_of_batches", None)
            leg["gpu_stage_sum_note"] = "SUM of the HIP-event stage times of groups that overlap on the device / wall: above 1 when they do"
        # the pure-Python restatement on a few reads
        k = min(a.py_reads, a.reads)
        cfg.args.max_reads = k
        py = {}
        if k > 0:
            t0 = time.perf_counter()
            rds = []
            pyb = bam.BamFile(bp) if a.reads <= 8000 else None
            if pyb is not None:
                refs = bam.read_fasta(fa)
                t1 = time.perf_counter()
                rds = list(bam.get_read_data(pyb, refs))
                bam.realign_reads(ctx, rds, os.path.join(tmp, "py.sam"), r=a.r)
                t2 = time.perf_counter()
                py = {"reads": k, "parse_whole_bam_s": round(t1 - t0, 3), "per_read_pipeline_ms": round((t2 - t1) / max(k, 1) * 1e3, 2)}
        line = {"metric": "BAM->SAM realigned reads/sec (end to end, file to file)", "value": one_pass["reads_per_s"], "unit": "reads/s",
                "value_is": "the one-pass reader (what `python -m npore_amd.realign` uses for one process and whole-contig regions)",
                "one_pass": one_pass, "one_pass_output_identical": same_one_pass,
                "reads": a.reads, "distinct_reads": a.reads if a.distinct <= 0 else min(a.distinct, a.reads),
                "qualities": "constant 20" if a.const_qual else "uniform per base over phred 0 ... 93 (reference test/generate_bam.py:63,79)",
                "ref_len": a.ref_len, "r": a.r, "batch": a.batch,
                "host_cpus": len(os.sched_getaffinity(0)), "bam_bytes": os.path.getsize(bp),
                "resident": resident, "streamed": streamed, "streamed_output_identical": same,
                "cold_start": cold,
                "rss_mb_before_timed_runs": round(rss0 / 1024.0, 1), "python_restatement": py, "input_generation_s": round(t_gen, 1)}
        print(json.dumps(line))
        ctx.close()


if __name__ == "__main__":
    main()
