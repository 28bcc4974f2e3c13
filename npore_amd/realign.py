#!/usr/bin/env python3
"""realign.py -- BAM in, SAM out, same flags as reference src/realign.py:15-71.

    python -m npore_amd.realign --bam reads.bam --ref ref.fasta --out_prefix out \\
                                [--stats_dir DIR] [--contig ...] [--max_reads N] ...

The per-read DP runs on an MI355X through libnpore_amd.so; reads are handed to
it in batches instead of one align() call per pool worker
(src/realign.py:110-114).  Several GPUs: launch one process per GPU with
`python -m torch.distributed.run --nproc-per-node N -m npore_amd.realign ...`;
rank k realigns reads k, k+N, ... into its own part file and rank 0 appends
the parts to the SAM (record order is arbitrary in the reference too).
--recalc_cms recounts the confusion matrices with `samtools mpileup` like the reference (bam.get_confusion_matrices); --plot is out of scope.
"""
import argparse
import os
import sys
from time import perf_counter

import numpy as np

from . import aln, bam as bam_mod, cfg, dist as dist_mod


def argparser():
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--bam", required=True, help="Input BAM to be realigned.")
    parser.add_argument("--ref", required=True, help="Input reference FASTA.")
    parser.add_argument("--out_prefix", required=True, help="Output SAM file prefix.")
    parser.add_argument("--contig", type=str, help="Single contig to realign (with --contig_beg/--contig_end).")
    parser.add_argument("--contig_beg", type=int, help="Start of realigned region.")
    parser.add_argument("--contig_end", type=int, help="End of realigned region.")
    parser.add_argument("--contigs", type=str, help="Comma-separated contigs to realign.")
    parser.add_argument("--max_reads", type=int, default=0, help="Limit on realigned reads (0 = all).")
    parser.add_argument("--bed", type=str, help="BED file of regions to realign.")
    parser.add_argument("--max_n", type=int, default=6, help="Maximum n-polymer period considered.")
    parser.add_argument("--max_l", type=int, default=100, help="Maximum n-polymer repeat count considered.")
    parser.add_argument("--chunk_width", type=int, default=100000, help="(confusion-matrix recalculation only)")
    parser.add_argument("--stats_dir", default=None,
                        help="Directory with subs/nps/inss/dels _cm.npy (default: the shipped guppy5_stats; with --recalc_cms the "
                             "recounted matrices are written here, default ./stats as in the reference).")
    parser.add_argument("--plot", action="store_true", help="(not supported in this build)")
    parser.add_argument("--recalc_cms", action="store_true",
                        help="Recount the confusion matrices from the BAM (needs `samtools mpileup`) instead of loading them.")
    parser.add_argument("--recalc_exit", action="store_true", help="Exit after --recalc_cms.")
    # additions
    parser.add_argument("--batch_reads", type=int, default=4000, help="Reads per GPU batch (4 000 reads of 10 kb fill the GPU once at the default band, Context.round_chunks; file to file 6 000 - 8 000 measured 5 - 7 %% faster at twice the device memory: 49 GB of traceback words per batch in flight, three in flight).")
    parser.add_argument("--device", type=int, default=int(os.environ.get("LOCAL_RANK", "0")), help="HIP device.")
    parser.add_argument("--python_io", action="store_true",
                        help="Use the pure-Python BAM reader / SAM writer (the restatement the native one is tested against).")
    return parser


def main():
    if cfg.args.plot:
        print("\nERROR: --plot is not available in npore_amd (matplotlib reports are out of scope).")
        sys.exit(1)
    native = not cfg.args.python_io
    # one process per GPU: the host stages of a rank use its share of the node's cores, and the BAM is inflated
    # once per node (local rank 0; the other ranks map its copy, bam.NativeBam)
    threads = dist_mod.host_threads_per_rank() if int(os.environ.get("LOCAL_WORLD_SIZE", "1")) > 1 else 0
    print("> reading reference")
    ref_seqs = bam_mod.NativeFasta(cfg.args.ref) if native else bam_mod.read_fasta(cfg.args.ref)
    print("> selecting BAM regions")
    # One process, the native reader, batches written by the library: ONE PASS over the BAM (bam.NativeBam.realign_sequential:
    # no record index, every block inflated once, ingest overlapped with the GPU) -- the way the reference itself reads,
    # a generator over bam.fetch() (src/bam.pyx:18-47).  Several ranks, or regions / a file order that rule it out: the
    # indexed reader.
    # Several ranks on one file: each walks a contiguous stretch of the record stream in one pass, cut at record starts taken
    # from the .bai linear index (bam.NativeBam.set_share) -- the counterpart of the reference feeding all its workers from
    # one sequential read (src/bam.pyx:18-47, src/realign.py:110-114); without a .bai, or with --max_reads (the ranks cannot
    # know how many reads the others keep), the indexed reader.  The choice depends on the files and flags alone: every
    # rank makes the same one.
    world_size = dist_mod.world()[1]
    one_pass = (native and cfg.args.batch_reads > 0 and not getattr(cfg.args, "bed", None)
                and os.environ.get("NPORE_BAM_ONE_PASS", "1") != "0" and bam_mod.NativeBam.is_bgzf(cfg.args.bam)
                and (world_size == 1 or (not cfg.args.max_reads and bam_mod.NativeBam.bai_path(cfg.args.bam) is not None)))
    bam = bam_mod.NativeBam(cfg.args.bam, threads=threads, one_pass=one_pass) if native else bam_mod.BamFile(cfg.args.bam)
    if one_pass and world_size > 1:
        try:
            bam.set_share(dist_mod.world()[0], world_size)
        except bam_mod.OnePassUnsupported as e:
            print(f"    ({e}: taking the indexed reader)")
            bam.close()
            one_pass = False
            bam = bam_mod.NativeBam(cfg.args.bam, threads=threads)
    bam_mod.get_bam_regions(bam, ref_seqs)

    if cfg.args.recalc_cms:              # src/realign.py:92-95 + src/bam.pyx:166-200
        # (the pileup counter slices the contigs: the native reader only serves lengths, so take the sequences too)
        cfg.args.refs = bam_mod.NativeFastaSeqs(cfg.args.ref) if native else ref_seqs
        subs, nps, inss, dels = bam_mod.get_confusion_matrices()
        print("> calculating score matrices")
        cfg.args.sub_scores, cfg.args.np_scores, cfg.args.ins_scores, cfg.args.del_scores = \
            aln.calc_score_matrices(subs, nps, inss, dels)
    else:
        print("> calculating score matrices")
        cfg.args.sub_scores, cfg.args.np_scores, cfg.args.ins_scores, cfg.args.del_scores = \
            aln.load_default_tables(cfg.args.stats_dir)

    rank, world, _ = dist_mod.world()
    if world > 1 and not native:
        print("\nERROR: --python_io runs on one GPU only.")
        sys.exit(1)
    print("> creating output SAM")
    final_sam = f"{cfg.args.out_prefix}.sam"
    out_sam = final_sam if world == 1 else f"{cfg.args.out_prefix}.part{rank}.sam"
    if rank == 0:
        bam_mod.create_header(final_sam, bam)
    if world > 1:
        open(out_sam, "w").close()

    print("> extracting read data from BAM")
    start = perf_counter()
    n_dev = max(aln.device_count(), 1)
    ctx = aln.Context(cfg.args.sub_scores, cfg.args.np_scores, device=cfg.args.device % n_dev)
    n = 0
    done = False
    if native and one_pass:
        print("> computing individual read realignments")
        header_bytes = os.path.getsize(out_sam)
        try:
            n, bad, _ = bam.realign_sequential(ctx, ref_seqs, cfg.args.regions, out_sam, batch_reads=cfg.args.batch_reads,
                                               max_reads=cfg.args.max_reads, threads=threads)
            for k, st in bad:
                print(f"\nERROR: read #{k} of the selected reads: " +
                      ("CIGAR does not match sequence lengths; skipped." if st & 32 else f"inconsistent traceback (status {st})"))
            done = True
        except bam_mod.OnePassUnsupported as e:
            if world > 1 and "regions" not in str(e) and "region per contig" not in str(e):
                # found in the DATA of one rank's stretch (a file that is not sorted): the other ranks may not have seen it,
                # and a rank that changed readers on its own would deal the reads differently from the rest
                print(f"\nERROR: {e}.")
                sys.exit(1)
            print(f"    ({e}: taking the indexed reader)")
            with open(out_sam, "r+b") as fh:
                fh.truncate(header_bytes)
            bam.close()
            bam = bam_mod.NativeBam(cfg.args.bam, threads=threads)
    if done:
        bam.close()
        ref_seqs.close()
        if world > 1:
            n = dist_mod.gather_parts(final_sam, cfg.args.out_prefix, n)
    elif native:
        idx = bam.select(cfg.args.regions, cfg.args.max_reads)
        # reads are independent: dealt by index, no data-path collective.  A resident BAM is dealt round-robin; a
        # STREAMED one in contiguous shares, so that a rank only ever inflates the blocks that hold its own reads
        if bam.streamed and world > 1:
            per = (len(idx) + world - 1) // world
            idx = idx[rank * per:(rank + 1) * per]
        else:
            idx = idx[rank::world]
        print("> computing individual read realignments")
        n += bam_mod.realign_native(ctx, bam, ref_seqs, idx, out_sam, batch_reads=cfg.args.batch_reads, threads=threads)
        bam.close()
        ref_seqs.close()
        if world > 1:
            n = dist_mod.gather_parts(final_sam, cfg.args.out_prefix, n)
    else:
        read_data = bam_mod.get_read_data(bam, ref_seqs)
        print("> computing individual read realignments")
        batch = []
        for rd in read_data:
            batch.append(rd)
            if len(batch) >= cfg.args.batch_reads:
                n += bam_mod.realign_reads(ctx, batch, out_sam)
                batch = []
        n += bam_mod.realign_reads(ctx, batch, out_sam)
    ctx.close()
    print(f"    {n} reads, runtime: {perf_counter() - start:.2f}s")


if __name__ == "__main__":
    cfg.args = argparser().parse_args()
    try:
        main()
    except KeyboardInterrupt:
        print("\nERROR: Program terminated.")
        sys.exit(1)
