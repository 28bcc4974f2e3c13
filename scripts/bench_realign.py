#!/usr/bin/env python3
"""End-to-end BAM -> SAM throughput of the realign path with a stage breakdown.

A synthetic BAM is made from `--distinct` reads of the bench generator laid end to end on one contig and repeated to
`--reads` records (the record block is replicated at the byte level and compressed on a thread pool, so a 48 000-read /
0.7 GB file takes seconds, not minutes).  One JSON line: the stages of the native pipeline (open = BGZF inflate + record
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


def build_inputs(tmp, n, distinct, ref_len, seed):
    distinct = min(distinct, n)
    refs, seqs, cigs = synth.make_batch(seed, distinct, ref_len=ref_len)
    dec = np.frombuffer(b"NACGT", np.uint8)
    contig, recs, pos = [], [], 0
    for k, (rf, sq, cg) in enumerate(zip(refs, seqs, cigs)):
        cg = np.frombuffer(cg, np.uint8)
        edges = np.flatnonzero(np.diff(cg)) + 1
        starts = np.concatenate(([0], edges)); lens = np.diff(np.concatenate((starts, [len(cg)])))
        ops = [("MIDNSHP=XB".index(chr(cg[s])), int(l)) for s, l in zip(starts, lens)]
        recs.append(dict(name=f"read{k}", flag=0, ref_id=0, pos=pos, cigar=ops, seq=dec[sq].tobytes().decode(),
                         qual=bytes([20]) * len(sq), hp=k % 3))
        contig.append(dec[rf].tobytes())
        pos += len(rf)
    contig = b"".join(contig)
    fa = os.path.join(tmp, "ref.fa")
    with open(fa, "wb") as fh:
        fh.write(b">ctg\n")
        for i in range(0, len(contig), 60):
            fh.write(contig[i:i + 60] + b"\n")
    small = os.path.join(tmp, "distinct.bam")
    bam.write_bam(small, [("ctg", len(contig))], recs, level=1)
    raw = bam._bgzf_decompress(small)
    l_text, = struct.unpack_from("<i", raw, 4)
    p = 8 + l_text
    n_ref, = struct.unpack_from("<i", raw, p); p += 4
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", raw, p); p += 8 + l_name
    header, body = raw[:p], raw[p:]
    reps, rest = divmod(n, distinct)
    # the first `rest` records once more
    q = 0
    for _ in range(rest):
        bs, = struct.unpack_from("<i", body, q); q += 4 + bs
    bp = os.path.join(tmp, "reads.bam")
    bgzf_write(bp, header + body * reps + body[:q])
    os.remove(small)
    return bp, fa, len(contig)


def run_file(ctx, bp, fa, clen, a, out, stream):
    """open -> select -> npore_bam_realign_file; returns the stage dictionary"""
    t0 = time.perf_counter()
    nb, nf = bam.NativeBam(bp, stream=stream, threads=a.threads), bam.NativeFasta(fa)
    t1 = time.perf_counter()
    idx = nb.select([("ctg", 0, clen - 1)])
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


def run_one_pass(ctx, bp, fa, clen, a, out):
    """header-only open -> npore_bam_realign_sequential (what `realign` does for one process and whole-contig regions)"""
    t0 = time.perf_counter()
    nb, nf = bam.NativeBam(bp, one_pass=True, threads=a.threads), bam.NativeFasta(fa)
    t1 = time.perf_counter()
    bam.create_header(out, nb)
    n, bad, _ = nb.realign_sequential(ctx, nf, [("ctg", 0, clen - 1)], out, batch_reads=a.batch, r=a.r, threads=a.threads)
    t2 = time.perf_counter()
    ft = nb.file_timing()
    res = {"one_pass": True, "reads": int(n), "open_header_s": round(t1 - t0, 3), "realign_s": round(t2 - t1, 3), "total_s": round(t2 - t0, 3),
           "reads_per_s": round(n / (t2 - t0), 1),
           "stage_sums_s": {k[:-3]: round(v * 1e-3, 3) for k, v in ft.items() if k not in ("wall_ms",)},
           "stage_sums_note": "fetch_pack includes the inflation of every block (once); align_call = waiting for a batch's completion event; "
                              "gpu_kernels / pcie are sums of stage times of groups that overlap on the device",
           "sam_bytes": os.path.getsize(out) if out != "/dev/null" else None, "bad_reads": len(bad)}
    nb.close(); nf.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=48000)
    ap.add_argument("--distinct", type=int, default=4000)
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
    a = ap.parse_args()
    sub, nps, _, _ = aln.load_default_tables()
    with tempfile.TemporaryDirectory(dir=a.tmp) as tmp:
        t = time.perf_counter()
        # (in a child process: the generator's 0.7 GB of Python byte strings must not count towards this process's peak RSS)
        import multiprocessing as mp
        with mp.get_context("spawn").Pool(1) as pool:
            bp, fa, clen = pool.apply(build_inputs, (tmp, a.reads, a.distinct, a.ref_len, a.seed))
        t_gen = time.perf_counter() - t
        ctx = aln.Context(sub, nps)
        cfg.args = argparse.Namespace(max_n=6, max_l=100, regions=[("ctg", 0, clen - 1)], max_reads=0)
        out = os.path.join(tmp, "out.sam")
        # the STREAMED handle first: ru_maxrss is the peak of the whole process so far, and the resident handle holds
        # the inflated file (the context's page-locked staging and the GPU runtime are in both figures)
        rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
        if a.one_pass_only:
            one_pass = run_one_pass(ctx, bp, fa, clen, a, "/dev/null")
            one_pass["peak_rss_mb"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, 1)
            print(json.dumps({"metric": "BAM->SAM realigned reads/sec (one-pass ingest, SAM text discarded)", "value": one_pass["reads_per_s"],
                              "unit": "reads/s", "reads": a.reads, "distinct_reads": min(a.distinct, a.reads), "r": a.r, "batch": a.batch,
                              "bam_bytes": os.path.getsize(bp), "one_pass": one_pass, "rss_mb_before_timed_runs": round(rss0 / 1024.0, 1),
                              "input_generation_s": round(t_gen, 1)}))
            ctx.close()
            return
        streamed = run_file(ctx, bp, fa, clen, a, "/dev/null" if a.streamed_only else out + ".s", True)
        streamed["peak_rss_mb"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, 1)
        if a.streamed_only:
            print(json.dumps({"metric": "BAM->SAM realigned reads/sec (streamed ingest, SAM text discarded)", "value": streamed["reads_per_s"],
                              "unit": "reads/s", "reads": a.reads, "distinct_reads": min(a.distinct, a.reads), "r": a.r, "batch": a.batch,
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
                "reads": a.reads, "distinct_reads": min(a.distinct, a.reads), "ref_len": a.ref_len, "r": a.r, "batch": a.batch,
                "host_cpus": len(os.sched_getaffinity(0)), "bam_bytes": os.path.getsize(bp),
                "resident": resident, "streamed": streamed, "streamed_output_identical": same,
                "rss_mb_before_timed_runs": round(rss0 / 1024.0, 1), "python_restatement": py, "input_generation_s": round(t_gen, 1)}
        print(json.dumps(line))
        ctx.close()


if __name__ == "__main__":
    main()
