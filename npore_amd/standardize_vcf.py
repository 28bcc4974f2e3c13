"""standardize_vcf: counterpart of reference src/standardize_vcf.py:10-92 -- rewrite the
variants of a phased VCF in nPoRe's standard representation: split into haplotypes, apply each to
the reference, realign every haplotype sequence against the reference with align() (GPU, all
sequences of both haplotypes in one batch), standardise, and turn the CIGARs back into one VCF.

Same CLI flags as the reference (the positional form of test/test_std_vcf.sh is not accepted by
the reference's own parser either).  Additions: --device, --r, --max_b_rows (the reference uses
align()'s defaults, r=30 / max_b_rows=20000, src/bam.pyx:102).

Usage:  python -m npore_amd.standardize_vcf --vcf in.vcf.gz --ref ref.fasta --out_prefix out
"""
import argparse
import os
import sys
from time import perf_counter

from . import aln, bam as bam_mod, cfg, vcf as vcf_mod


def argparser():
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--vcf", type=str, required=True, help="Input VCF to standardize.")
    parser.add_argument("--ref", type=str, required=True, help="Input reference FASTA corresponding to VCF.")
    parser.add_argument("--out_prefix", type=str, required=True, help="Output VCF prefix.")
    parser.add_argument("--contig", type=str, help="Single contig to standardize (with --contig_beg/--contig_end).")
    parser.add_argument("--contig_beg", type=int, help="Start of standardized region.")
    parser.add_argument("--contig_end", type=int, help="End of standardized region.")
    parser.add_argument("--contigs", type=str, help="Comma-separated contigs to standardize.")
    parser.add_argument("--stats_dir", default=None,
                        help="Directory with subs/nps/inss/dels _cm.npy (default: the shipped guppy5_stats).")
    parser.add_argument("--max_n", type=int, default=6, help="Maximum n-polymer period considered.")
    parser.add_argument("--max_l", type=int, default=100, help="Maximum n-polymer repeat count considered.")
    parser.add_argument("--chunk_width", type=int, default=100000, help="(confusion-matrix recalculation only)")
    parser.add_argument("--min_qual", type=int, default=0, help="Only apply variants with quality above this threshold.")
    # additions
    parser.add_argument("--device", type=int, default=int(os.environ.get("LOCAL_RANK", "0")), help="HIP device.")
    parser.add_argument("--r", type=int, default=30, help="Band half-width of the realignment.")
    parser.add_argument("--max_b_rows", type=int, default=20000, help="Anti-diagonals per independent chunk.")
    return parser


def standardize(vcf, ref_seqs, regions, ctx, out_prefix=None, min_qual=0, r=30, max_b_rows=20000):
    """The pipeline of src/standardize_vcf.py:22-41 on in-memory records.  Returns
    (merged records, hap1 data, hap2 data); with out_prefix also writes the files the reference
    writes ({prefix}pre1/2.vcf.gz, {prefix}1/2.vcf.gz, {prefix}.vcf.gz)."""
    recs1, recs2 = vcf_mod.split_vcf(vcf, regions)
    if out_prefix:
        for k, recs in ((1, recs1), (2, recs2)):
            vcf_mod.write_vcf(f"{out_prefix}pre{k}.vcf.gz", vcf.header, recs, vcf.samples or ("SAMPLE",))
    hap1 = vcf_mod.apply_vcf(recs1, 1, ref_seqs, regions, min_qual)
    hap2 = vcf_mod.apply_vcf(recs2, 2, ref_seqs, regions, min_qual)
    data = bam_mod.realign_haps(ctx, hap1 + hap2, r=r, max_b_rows=max_b_rows)
    hap1 = [x for x in data if x[1] == 1]
    hap2 = [x for x in data if x[1] == 2]
    out1, out2 = vcf_mod.gen_records(hap1), vcf_mod.gen_records(hap2)
    merged = vcf_mod.merge_records(out1, out2, regions)
    if out_prefix:
        vcf_mod.write_vcf(f"{out_prefix}1.vcf.gz", vcf_mod.gen_header(hap1), out1)
        vcf_mod.write_vcf(f"{out_prefix}2.vcf.gz", vcf_mod.gen_header(hap2), out2)
        vcf_mod.write_vcf(f"{out_prefix}.vcf.gz", vcf_mod.gen_header(hap1), merged)
    return merged, hap1, hap2


def main():
    start = perf_counter()
    print("> selecting vcf regions")
    ref_seqs = bam_mod.NativeFastaSeqs(cfg.args.ref)
    vcf = vcf_mod.VcfFile(cfg.args.vcf)
    vcf_mod.get_vcf_regions(ref_seqs, vcf)

    print("> calculating score matrices")
    cfg.args.sub_scores, cfg.args.np_scores, cfg.args.ins_scores, cfg.args.del_scores = \
        aln.load_default_tables(cfg.args.stats_dir)

    n_dev = max(aln.device_count(), 1)
    ctx = aln.Context(cfg.args.sub_scores, cfg.args.np_scores, device=cfg.args.device % n_dev)
    print("> splitting vcf, converting vcfs and ref to sequences, realigning hap sequences")
    merged, hap1, hap2 = standardize(vcf, ref_seqs, cfg.args.regions, ctx, cfg.args.out_prefix, cfg.args.min_qual,
                                     cfg.args.r, cfg.args.max_b_rows)
    ctx.close()
    print(f"    {len(hap1) + len(hap2)} haplotype sequences, {len(merged)} variants -> {cfg.args.out_prefix}.vcf.gz, "
          f"runtime: {perf_counter() - start:.2f}s")


if __name__ == "__main__":
    cfg.args = argparser().parse_args()
    cfg.args.recalc_cms = False
    try:
        main()
    except KeyboardInterrupt:
        print("\nERROR: Program terminated.")
        sys.exit(1)
