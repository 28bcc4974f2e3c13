#!/usr/bin/env python3
"""End-to-end BAM -> SAM throughput of the realign path on a synthetic BAM:
reads of the bench generator laid end to end on one contig.  Times the native host I/O pipeline
(libnpore_amd: inflate/index, select, pack + GPU align + standardise + format per batch, file write) and,
on a subset, the pure-Python restatement of the same steps.

    python scripts/bench_realign.py [--reads 4000] [--ref-len 10000] [--r 30] [--batch 2000] [--py-reads 64]
"""
import argparse, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from npore_amd import aln, bam, cfg, synth


def build_inputs(tmp, n, ref_len, seed):
    refs, seqs, cigs = synth.make_batch(seed, n, ref_len=ref_len)
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
    bp = os.path.join(tmp, "reads.bam")
    bam.write_bam(bp, [("ctg", len(contig))], recs, level=1)
    return bp, fa, len(contig)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=4000)
    ap.add_argument("--ref-len", type=int, default=10000)
    ap.add_argument("--r", type=int, default=30)
    ap.add_argument("--batch", type=int, default=2000)
    ap.add_argument("--py-reads", type=int, default=64)
    ap.add_argument("--seed", type=int, default=3)
    a = ap.parse_args()
    sub, nps, _, _ = aln.load_default_tables()
    ctx = aln.Context(sub, nps)
    with tempfile.TemporaryDirectory() as tmp:
        t = time.perf_counter()
        bp, fa, clen = build_inputs(tmp, a.reads, a.ref_len, a.seed)
        t_gen = time.perf_counter() - t
        cfg.args = argparse.Namespace(max_n=6, max_l=100, regions=[("ctg", 0, clen - 1)], max_reads=0)
        out = os.path.join(tmp, "out.sam")
        stages = {}
        for rep in range(2):          # rep 0 warms the GPU context and the page cache
            t0 = time.perf_counter()
            nb, nf = bam.NativeBam(bp), bam.NativeFasta(fa)
            t1 = time.perf_counter()
            idx = nb.select(cfg.args.regions)
            t2 = time.perf_counter()
            bam.create_header(out, nb)
            t_gpu = 0.0
            host = {}
            for k in range(0, len(idx), a.batch):
                bam.realign_native(ctx, nb, nf, idx[k:k + a.batch], out, r=a.r)
                for kk, v in nb.timing().items():
                    host[kk] = round(host.get(kk, 0.0) + v * 1e-3, 4)
                tm = ctx.timing()
                t_gpu += (tm["dev_prep_ms"] + tm["fill_ms"] + tm["traceback_ms"] + tm["h2d_ms"] + tm["d2h_ms"]) * 1e-3
            t3 = time.perf_counter()
            # the same through the library's overlapped batch loop
            bam.create_header(out + ".pipe", nb)
            bam.realign_native(ctx, nb, nf, idx, out + ".pipe", r=a.r, batch_reads=a.batch)
            t4 = time.perf_counter()
            same = open(out, "rb").read() == open(out + ".pipe", "rb").read()
            nb.close(); nf.close()
            stages = {"open_inflate_index_s": round(t1 - t0, 4), "select_s": round(t2 - t1, 4),
                      "batches_s": round(t3 - t2, 4), "of_which_gpu_and_pcie_s": round(t_gpu, 4), "library_stages_s": host, "total_s": round(t3 - t0, 4),
                      "overlapped_batches_s": round(t4 - t3, 4), "overlapped_total_s": round(t2 - t0 + t4 - t3, 4),
                      "overlapped_output_identical": same}
        native_rps = len(idx) / stages["overlapped_total_s"]
        # the pure-Python restatement on a subset
        k = min(a.py_reads, len(idx))
        cfg.args.max_reads = k
        t0 = t1 = t2 = time.perf_counter()
        if k > 0:
            py = bam.BamFile(bp)
            refs = bam.read_fasta(fa)
            t1 = time.perf_counter()
            rds = list(bam.get_read_data(py, refs))
            bam.realign_reads(ctx, rds, os.path.join(tmp, "py.sam"), r=a.r)
            t2 = time.perf_counter()
        line = {"metric": "BAM->SAM realigned reads/sec (end to end, file to file)", "value": round(native_rps, 1), "unit": "reads/s",
                "reads": int(len(idx)), "ref_len": a.ref_len, "r": a.r, "batch": a.batch, "host_cores": os.cpu_count(),
                "native": stages,
                "python_restatement": {"reads": k, "parse_whole_bam_s": round(t1 - t0, 3), "per_read_pipeline_ms": round((t2 - t1) / max(k, 1) * 1e3, 2)},
                "input_generation_s": round(t_gen, 1)}
        print(json.dumps(line))
    ctx.close()


if __name__ == "__main__":
    main()
