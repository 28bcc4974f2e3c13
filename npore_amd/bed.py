"""n-polymer BED files of a reference: counterpart of reference src/bed.py:15-177.

For every region of --bed (or --contig/--contigs) the reference is cut into --chunk_width slices
(src/bam.pyx:149-162), each annotated on its own with get_np_info (so a repeat crossing a slice border
is seen as two), and the n-polymer starts become regions [pos, pos + n*L).  Here all slices go through
the GPU in batches (npore_np_regions); what the reference then pipes through `bedtools merge`,
`sort -k1,1n -k2,2n -k3,3n`, `sed` and `bedtools complement -L` (src/bed.py:80-140) is done in numpy,
including the quirks of that pipeline: contig names lose a leading "chr" for sorting and get it back
only if they then start with a digit; the sort key of a non-numeric name is 0.

Outputs, as the reference: {out_prefix}_{n}.bed for n = 1..max_n (slop 1, merged), {out_prefix}_all.bed,
{out_prefix}_0.bed (complement of _all within the lengths given by column 3 of --bed), and the
`.genome` file next to --bed.

Usage:  python -m npore_amd.bed --ref ref.fasta --bed regions.bed --out_prefix out
"""
import argparse
import os
import re
import sys
from time import perf_counter

import numpy as np

from . import aln, bam as bam_mod, cfg
from .cig import bases_to_int


def argparser():
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--ref", required=True, help="Input reference FASTA.")
    parser.add_argument("--bed", required=True,
                        help="Input BED of regions for which to compute n-polymer information (also needed with "
                             "--contig* to find the complement).")
    parser.add_argument("--contig", type=str, help="Single contig (with --contig_beg/--contig_end).")
    parser.add_argument("--contig_beg", type=int, help="Start of region.")
    parser.add_argument("--contig_end", type=int, help="End of region.")
    parser.add_argument("--contigs", type=str, help="Comma-separated contigs.")
    parser.add_argument("-chunk_width", "--chunk_width", dest="chunk_width", type=int, default=1000000,
                        help="Reference is considered in slices of this size.")
    parser.add_argument("--max_n", type=int, default=6, help="Maximum n-polymer period considered.")
    parser.add_argument("--max_l", type=int, default=100, help="Maximum n-polymer repeat count considered.")
    parser.add_argument("--out_prefix", required=True, help="Output BED file prefix.")
    # additions
    parser.add_argument("--device", type=int, default=int(os.environ.get("LOCAL_RANK", "0")), help="HIP device.")
    parser.add_argument("--batch_bases", type=int, default=1 << 30, help="Reference bases per GPU batch.")
    return parser


def get_regions(ref_seqs):
    """cfg.args.regions as bed.py gets them from get_bam_regions (src/util.py:16-93 without a BAM):
    --contig / --contigs / the lines of --bed."""
    a = cfg.args
    if a.contig:
        if a.contig not in ref_seqs:
            print(f"ERROR: contig '{a.contig}' not present in '{a.ref}'. Valid contigs are: {list(ref_seqs)}")
            sys.exit(1)
        if a.contigs:
            print("\nERROR: can't set 'contig' and 'contigs'.")
            sys.exit(1)
        max_end = len(ref_seqs[a.contig]) - 1
        beg = a.contig_beg if a.contig_beg else 0
        end = a.contig_end if a.contig_end else max_end
        a.regions = [(a.contig, beg, min(max_end, end))]
    elif a.contigs:
        if a.contig_beg or a.contig_end:
            print("\nERROR: can't set start/endpoints with multiple contigs.")
            sys.exit(1)
        a.regions = []
        for contig in a.contigs.split(","):
            if contig not in ref_seqs:
                print(f"ERROR: contig '{contig}' not present in '{a.ref}'. Valid contigs are: {list(ref_seqs)}")
                sys.exit(1)
            a.regions.append((contig, 0, len(ref_seqs[contig]) - 1))
    else:
        try:
            with open(a.bed) as fh:
                a.regions = [(f[0], int(f[1]), int(f[2])) for f in (x.split() for x in fh) if f]
        except FileNotFoundError:
            print("\nERROR: could not open 'cfg.args.bed' BED.")
            sys.exit(1)
    return a.regions


def get_ranges(regions, chunk_width):
    """src/bam.pyx:149-162."""
    out = []
    for contig, start, stop in regions:
        for st in range(start, stop, chunk_width):
            out.append((contig, st, min(stop, st + chunk_width)))
    return out


def np_regions_of_ranges(ctx, ref_seqs, ranges, batch_bases=1 << 30):
    """get_np_regions (src/bed.py:56-76) over all ranges: per period index, (contig index array, start, stop)
    in range order.  contig index = position in the returned name list."""
    names = []
    index = {}
    per_n = [([], [], []) for _ in range(ctx.max_n)]
    batch, size = [], 0

    def flush():
        nonlocal batch, size
        if not batch:
            return
        res = ctx.np_regions([bases_to_int(ref_seqs[c][s:e]) for c, s, e in batch])
        for n in range(ctx.max_n):
            for (c, s, _e), (pos, reps) in zip(batch, res[n]):
                if len(pos):
                    start = pos.astype(np.int64) + s
                    per_n[n][0].append(np.full(len(pos), index[c], np.int64))
                    per_n[n][1].append(start)
                    per_n[n][2].append(start + (n + 1) * reps.astype(np.int64))
        batch, size = [], 0

    for c, s, e in ranges:
        if c not in index:
            index[c] = len(names)
            names.append(c)
        if size and size + (e - s) > batch_bases:
            flush()
        batch.append((c, s, e))
        size += e - s
    flush()
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.int64)
    return names, [(cat(a), cat(b), cat(c)) for a, b, c in per_n]


# ---------------------------------------------------------------------------
# the text pipeline of save_np_region_beds (src/bed.py:80-140) on arrays
def bed_merge(ctg, start, stop):
    """`bedtools merge` of intervals already grouped by contig and sorted by start: overlapping and
    book-ended intervals (start <= running end) become one."""
    if len(ctg) == 0:
        return ctg, start, stop
    # running maximum of stop within a block of lines of one contig: blocks are kept apart by an offset
    big = int(stop.max()) + 2
    block = np.zeros(len(ctg), np.int64)
    block[1:] = np.cumsum(ctg[1:] != ctg[:-1])
    key = block * big
    run = np.maximum.accumulate(stop + key)
    new = np.ones(len(ctg), bool)
    new[1:] = (ctg[1:] != ctg[:-1]) | (start[1:] + key[1:] > run[:-1])
    first = np.flatnonzero(new)
    last = np.concatenate((first[1:], [len(ctg)])) - 1
    return ctg[first], start[first], (run - key)[last]


_LEADING_INT = re.compile(r"\s*-?\d+")


def sort_names(names):
    """What `sed s/^chr// | sort -k1,1n ... | sed 's/^[0-9]/chr&/'` does to contig names: (numeric key, printed name)."""
    out = []
    for nm in names:
        bare = nm[3:] if nm.startswith("chr") else nm
        m = _LEADING_INT.match(bare)
        out.append((int(m.group()) if m else 0, ("chr" + bare) if bare[:1].isdigit() else bare))
    return out


def bed_sort(names, ctg, start, stop):
    """`sort -k1,1n -k2,2n -k3,3n` on the chr-stripped lines.  One deliberate difference: contigs whose numeric
    key is equal (non-numeric names all have key 0, "1" and "1_alt" both 1) are kept apart, ordered bytewise by
    name -- GNU sort would interleave their lines by start, which `bedtools merge` / `complement` downstream
    cannot digest.  For chr1..chr22-style names the result is the reference's.
    Returns the printed contig name of every contig and the permuted columns."""
    keyed = sort_names(names)
    knum = np.array([k for k, _ in keyed], np.int64)[ctg] if len(ctg) else np.zeros(0, np.int64)
    bare = [nm[3:] if nm.startswith("chr") else nm for nm in names]
    rank = np.argsort(np.argsort(np.array([b.encode() for b in bare], dtype=object), kind="stable"), kind="stable") \
        if names else np.zeros(0, np.int64)
    order = np.lexsort((stop, start, rank[ctg] if len(ctg) else ctg, knum))
    printed = [p for _, p in keyed]
    return printed, ctg[order], start[order], stop[order]


def write_bed(path, printed, ctg, start, stop):
    with open(path, "w") as fh:
        fh.write("".join(f"{printed[c]}\t{s}\t{e}\n" for c, s, e in zip(ctg.tolist(), start.tolist(), stop.tolist())))


def bed_complement(printed, ctg, start, stop, genome):
    """`bedtools complement -L -i all.bed -g genome`: the parts of every contig that has records which no record
    covers, contigs in the order of the (sorted, merged) input; genome: {printed name: length}."""
    oc, os_, oe = [], [], []
    i, n = 0, len(ctg)
    while i < n:
        c = int(ctg[i])
        j = i
        while j < n and ctg[j] == c:
            j += 1
        length = genome.get(printed[c])
        if length is None:
            print(f"ERROR: contig '{printed[c]}' of the n-polymer regions is not in the .genome file.")
            sys.exit(1)
        prev = 0
        for k in range(i, j):
            s, e = int(start[k]), int(stop[k])
            if s > prev:
                oc.append(c); os_.append(prev); oe.append(min(s, length))
            prev = max(prev, e)
        if prev < length:
            oc.append(c); os_.append(prev); oe.append(length)
        i = j
    a = lambda x: np.array(x, np.int64)
    return a(oc), a(os_), a(oe)


def save_np_region_beds(names, per_n, out_prefix, bed_path, slop=1):
    """src/bed.py:80-140."""
    max_n = len(per_n)
    sorted_n = []
    for n in range(1, max_n + 1):
        ctg, start, stop = per_n[n - 1]
        ctg, start, stop = bed_merge(ctg, np.maximum(0, start - slop), stop + slop)
        printed, ctg, start, stop = bed_sort(names, ctg, start, stop)
        write_bed(f"{out_prefix}_{n}.bed", printed, ctg, start, stop)
        sorted_n.append((ctg, start, stop))
    printed = [p for _, p in sort_names(names)]
    # cat | sed | sort | sed | bedtools merge: the lines now carry the printed names
    pidx = {}
    pnames = []
    for p in printed:
        if p not in pidx:
            pidx[p] = len(pnames)
            pnames.append(p)
    remap = np.array([pidx[p] for p in printed], np.int64) if printed else np.zeros(0, np.int64)
    cat = lambda k: np.concatenate([x[k] for x in sorted_n]) if sorted_n else np.zeros(0, np.int64)
    ctg, start, stop = cat(0), cat(1), cat(2)
    ctg = remap[ctg] if len(ctg) else ctg
    printed2, ctg, start, stop = bed_sort(pnames, ctg, start, stop)
    ctg, start, stop = bed_merge(ctg, start, stop)
    write_bed(f"{out_prefix}_all.bed", printed2, ctg, start, stop)

    if not bed_path:
        print("ERROR: 'cfg.args.bed' must be supplied.")
        sys.exit(1)
    if bed_path[-4:] != ".bed":
        print("ERROR: 'cfg.args.bed' is not BED file.")
        sys.exit(1)
    genome_path = f"{bed_path[:-4]}.genome"
    genome = {}
    with open(bed_path) as src, open(genome_path, "w") as dst:      # cut -f1,3
        for line in src:
            f = line.rstrip("\n").split("\t")
            dst.write("\t".join(f[0:1] + f[2:3]) + "\n")
            if len(f) >= 3:
                genome[f[0]] = int(f[2])
    cc, cs, ce = bed_complement(printed2, ctg, start, stop, genome)
    write_bed(f"{out_prefix}_0.bed", printed2, cc, cs, ce)
    return genome_path


def main():
    print("> extracting reference contigs")
    start = perf_counter()
    ref_seqs = bam_mod.NativeFastaSeqs(cfg.args.ref)
    get_regions(ref_seqs)
    for ctg, _s, _e in cfg.args.regions:
        if ctg not in ref_seqs:
            print(f"ERROR: contig '{ctg}' not present in '{cfg.args.ref}'.")
            sys.exit(1)
    print("> subdividing into chunks")
    ranges = get_ranges(cfg.args.regions, cfg.args.chunk_width)
    print(f"> computing repeat BEDs, n = 1-{cfg.args.max_n}")
    n_dev = max(aln.device_count(), 1)
    # a context wants penalty tables; the annotation does not use them
    ctx = aln.Context(np.zeros((5, 5), np.float32), np.zeros((cfg.args.max_n, cfg.args.max_l + 1, cfg.args.max_l + 1), np.float32),
                      max_n=cfg.args.max_n, max_l=cfg.args.max_l, device=cfg.args.device % n_dev)
    names, per_n = np_regions_of_ranges(ctx, ref_seqs, ranges, cfg.args.batch_bases)
    ctx.close()
    print(f"    runtime: {perf_counter() - start:.2f}s")
    print(f"> saving n-polymer BEDs, n = 1-{cfg.args.max_n}")
    start = perf_counter()
    save_np_region_beds(names, per_n, cfg.args.out_prefix, cfg.args.bed)
    print(f"    runtime: {perf_counter() - start:.2f}s")


if __name__ == "__main__":
    cfg.args = argparser().parse_args()
    try:
        main()
    except KeyboardInterrupt:
        print("\nERROR: Program terminated.")
        sys.exit(1)
