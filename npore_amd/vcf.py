"""VCF in / VCF out around realign_haps(): counterparts of reference src/vcf.py
(split_vcf :37-137, merge_vcfs :141-210, apply_vcf :214-272, gen_vcf :276-426) and
src/util.py:97-150 (get_vcf_regions), for the standardize_vcf driver -- the second
caller of align() (whole haplotype sequences, thousands of independent chunks each).

pysam is not needed: a VCF is tab-separated text, optionally BGZF/gzip compressed.
Records are kept in memory between the stages; the intermediate per-haplotype files
the reference leaves behind ({prefix}pre1/2.vcf.gz, {prefix}1/2.vcf.gz) are written
too.  No tabix index is made unless a `tabix` binary is on PATH (the reference calls
it through subprocess).

The CIGAR -> variant conversion works on whole chromosomes, so it is vectorised
(numpy) instead of the reference's per-character loop; tests/ hold the literal
per-character restatement it is checked against.
"""
import gzip
import os
import shutil
import struct
import subprocess
import sys
import zlib

import numpy as np

from . import cfg


class VcfRecord:
    """One data line.  pos is 1-based (pysam's .pos); start = pos-1; stop = start + len(REF)."""
    __slots__ = ("contig", "pos", "id", "alleles", "qual", "filter", "info", "format", "samples", "gt")

    def __init__(self, contig, pos, alleles, qual=None, id=".", filter="PASS", info=".", format=None, samples=(), gt=()):
        self.contig, self.pos, self.id, self.alleles = contig, int(pos), id, tuple(alleles)
        self.qual, self.filter, self.info, self.format = qual, filter, info, format
        self.samples, self.gt = tuple(samples), tuple(gt)

    @property
    def start(self):
        return self.pos - 1

    @property
    def stop(self):
        return self.pos - 1 + len(self.alleles[0])

    def key(self):
        return (self.contig, self.pos, self.alleles)


def _parse_gt(fmt, sample):
    """GT of the first... of ONE sample column as a tuple of allele indices (None for '.')."""
    if not fmt:
        return ()
    keys = fmt.split(":")
    if "GT" not in keys:
        return ()
    vals = sample.split(":")
    k = keys.index("GT")
    if k >= len(vals):
        return ()
    return tuple(None if a in (".", "") else int(a) for a in vals[k].replace("|", "/").split("/"))


class VcfFile:
    """A whole VCF in memory: header lines, sample names, records per contig in file order."""

    def __init__(self, path):
        try:
            raw = open(path, "rb").read()
        except (IOError, OSError):
            print(f"\nERROR: could not open VCF '{path}'.")          # src/util.py:105-109
            sys.exit(1)
        if raw[:2] == b"\x1f\x8b":
            raw = gzip.decompress(raw)                               # BGZF = concatenated gzip members
        self.header, self.samples, self.by_contig = [], [], {}
        for line in raw.decode().splitlines():
            if not line:
                continue
            if line.startswith("##"):
                self.header.append(line)
                continue
            f = line.split("\t")
            if line.startswith("#"):
                self.samples = f[9:]
                continue
            if len(f) < 8:
                print(f"\nERROR: malformed VCF line in '{path}': {line[:60]}")
                sys.exit(1)
            alts = [] if f[4] == "." else f[4].split(",")
            fmt = f[8] if len(f) > 8 else None
            smp = f[9:]
            # "only deal with 1-sample VCFs for now": the LAST sample's GT is the one used (src/vcf.py:56-57)
            gt = _parse_gt(fmt, smp[-1]) if smp else ()
            rec = VcfRecord(f[0], f[1], [f[3]] + alts, None if f[5] == "." else float(f[5]), f[2], f[6], f[7], fmt, smp, gt)
            self.by_contig.setdefault(f[0], []).append(rec)

    @property
    def contigs(self):
        return list(self.by_contig)

    @property
    def header_contigs(self):
        """IDs of the ##contig header lines, in header order (contigs that only appear in records after them)."""
        ids = []
        for h in self.header:
            if h.startswith("##contig=<"):
                for part in h[len("##contig=<"):].rstrip(">").split(","):
                    if part.startswith("ID="):
                        ids.append(part[3:])
        return ids + [c for c in self.by_contig if c not in ids]

    def fetch(self, contig, start, stop):
        """Records overlapping the half-open interval [start, stop), in file order."""
        return [r for r in self.by_contig.get(contig, ()) if r.start < stop and r.stop > start]


def fetch(records, contig, start, stop):
    return [r for r in records if r.contig == contig and r.start < stop and r.stop > start]


def get_vcf_regions(ref_seqs, vcf):
    """cfg.args.regions from --contig / --contigs / everything (src/util.py:97-150).
    ref_seqs: {contig: sequence}; vcf: VcfFile."""
    a = cfg.args
    if getattr(a, "contig", None):
        if getattr(a, "contigs", None):
            print("\nERROR: can't set 'contig' and 'contigs'.")
            sys.exit(1)
        if a.contig not in ref_seqs:
            print(f"\nERROR: contig '{a.contig}' not in FASTA.")
            sys.exit(1)
        if not a.contig_beg:
            a.contig_beg = 0
        if not a.contig_end:
            a.contig_end = len(ref_seqs[a.contig]) - 1
        a.regions = [(a.contig, a.contig_beg, a.contig_end)]
    elif getattr(a, "contigs", None):
        if a.contig_beg or a.contig_end:
            print("\nERROR: can't set start/endpoints with multiple contigs.")
            sys.exit(1)
        a.regions = []
        for contig in a.contigs.split(","):
            if contig not in ref_seqs:
                print(f"\nERROR: contig '{contig}' not in FASTA.")
                sys.exit(1)
            a.regions.append((contig, 0, len(ref_seqs[contig]) - 1))
    else:
        if getattr(a, "contig_beg", None) or getattr(a, "contig_end", None):
            print("\nERROR: 'contig' not supplied, but start/endpoints set.")
            sys.exit(1)
        a.regions = []
        for ctg in vcf.header_contigs:                                 # src/util.py:142-154
            if ctg not in ref_seqs:
                print(f"WARNING: contig '{ctg}' present in '{getattr(a, 'vcf', 'VCF')}', but"
                      f" not '{getattr(a, 'ref', 'FASTA')}', skipping...")
                continue
            l = len(ref_seqs[ctg])
            if vcf.fetch(ctg, 0, l - 1):                               # only contigs with variants
                a.regions.append((ctg, 0, l - 1))
    return a.regions


def _hap_record(rec, alleles=None):
    return VcfRecord(rec.contig, rec.pos, alleles if alleles is not None else rec.alleles, rec.qual, rec.id,
                     rec.filter, rec.info, rec.format, rec.samples, ())


def split_vcf(vcf, regions, filter_unphased=False):
    """Phased diploid records -> (hap1 records, hap2 records), src/vcf.py:37-137."""
    out1, out2 = [], []
    unphased, records = True, False
    for ctg, start, stop in regions:
        for rec in vcf.fetch(ctg, start, stop):
            records = True
            gt = tuple(0 if g is None else g for g in rec.gt) if rec.gt else (0, 0)
            if len(gt) == 1:
                gt = (gt[0], gt[0])
            if len(rec.alleles) == 3:                                   # two different variants
                if rec.alleles[gt[0]] != "*":                           # (spanning deletions skipped)
                    out1.append(_hap_record(rec, (rec.alleles[0], rec.alleles[gt[0]])))
                if rec.alleles[gt[1]] != "*":
                    out2.append(_hap_record(rec, (rec.alleles[0], rec.alleles[gt[1]])))
            elif gt[0] and gt[1]:
                out1.append(_hap_record(rec))
                out2.append(_hap_record(rec))
            elif gt[0]:
                if filter_unphased and not (rec.format and "PS" in rec.format.split(":")):
                    continue
                out1.append(_hap_record(rec))
            elif gt[1]:
                if filter_unphased and not (rec.format and "PS" in rec.format.split(":")):
                    continue
                out2.append(_hap_record(rec))
            elif len(rec.alleles) > 1 and rec.alleles[0] == rec.alleles[1]:
                pass                                                    # same as the reference base
            else:                                                       # treated as a homozygous variant
                out1.append(_hap_record(rec))
                out2.append(_hap_record(rec))
            if gt[0] and not gt[1]:
                unphased = False
    if not records:
        print("\nWARNING: VCF file has no variants in selected region.")
    elif unphased:
        print("\nWARNING: VCF file may be unphased.")
    return out1, out2


def apply_vcf(records, hap, ref_seqs, regions, min_qual=0):
    """Haplotype sequence + its edit script against the reference for every region (src/vcf.py:214-272):
    [(contig, hap, seq, ref, cigar over '=XID')]."""
    data = []
    for contig, start, stop in regions:
        ref = ref_seqs[contig]
        len_ref = len(ref)
        cig, seq = [], []
        ref_ptr = 0
        for rec in fetch(records, contig, start, stop):
            if len(rec.alleles) < 2:
                continue
            pos = rec.pos - 1
            if (min_qual and not rec.qual) or (rec.qual and rec.qual < min_qual):
                continue
            a0, a1 = rec.alleles[0], rec.alleles[1]
            indel_len = len(a1) - len(a0)
            if pos < ref_ptr:                                           # overlaps the previous deletion
                if indel_len > 0:                                       # insertions are let through
                    seq.append(a1[len(a0):])
                    cig.append("I" * indel_len)
                elif indel_len < 0 and pos == ref_ptr - 1:              # only its anchor base overlaps
                    cig.append("D" * -indel_len)
                    ref_ptr += -indel_len
                continue
            seq.append(ref[ref_ptr:pos])
            cig.append("=" * (pos - ref_ptr))
            ref_ptr = pos
            seq.append(a1)
            minlen = min(len(a0), len(a1))
            cig.append("".join("=" if a0[i] == a1[i] else "X" for i in range(minlen)))
            ref_ptr += minlen
            if indel_len > 0:
                cig.append("I" * indel_len)
            elif indel_len < 0:
                cig.append("D" * -indel_len)
                ref_ptr += -indel_len
        cig.append("=" * (len_ref - ref_ptr))
        seq.append(ref[ref_ptr:])
        data.append((contig, hap, "".join(seq), ref, "".join(cig)))
    return data


def gen_records(hap_data):
    """Expanded CIGAR of every (contig, hap, seq, ref, cigar) -> variant records (src/vcf.py:300-378):
    'X' and mismatching 'M' -> one substitution each; a run of 'D' / 'I' -> one record anchored on the
    previous reference base (none at reference position 0 / sequence position 0).  QUAL 60, FILTER PASS."""
    out = []
    for contig, _hap, seq, ref, cigar in hap_data:
        ops = np.frombuffer(cigar.encode(), dtype=np.uint8)
        n = len(ops)
        if n == 0:
            continue
        bad = ~np.isin(ops, np.frombuffer(b"=XMDI", dtype=np.uint8))
        if bad.any():
            print(f"\nERROR: unrecognized CIGAR operation '{chr(ops[np.argmax(bad)])}'")
            sys.exit(1)
        is_d, is_i = ops == ord("D"), ops == ord("I")
        # reference / sequence positions BEFORE each op
        ref_ptr = np.zeros(n + 1, np.int64); np.cumsum(~is_i, out=ref_ptr[1:])
        seq_ptr = np.zeros(n + 1, np.int64); np.cumsum(~is_d, out=seq_ptr[1:])
        rb = np.frombuffer(ref.encode(), dtype=np.uint8)
        sb = np.frombuffer(seq.encode(), dtype=np.uint8)
        # substitutions
        diag = np.flatnonzero(~is_d & ~is_i)
        rp, sp = ref_ptr[diag], seq_ptr[diag]
        sub = (ops[diag] == ord("X")) | ((ops[diag] == ord("M")) & (rb[rp] != sb[sp]))
        events = [(int(p), 0, int(q), 1) for p, q in zip(rp[sub], sp[sub])]
        # indel runs (a run ends where the op changes)
        for flag, kind in ((is_d, 1), (is_i, 2)):
            f = flag.astype(np.int8)
            starts = np.flatnonzero(np.diff(np.concatenate(([0], f))) == 1)
            ends = np.flatnonzero(np.diff(np.concatenate((f, [0]))) == -1) + 1
            for s, e in zip(starts, ends):
                events.append((int(ref_ptr[s]), kind, int(seq_ptr[s]), int(e - s)))
        # file order = CIGAR order: by op index, which the pair (ref position, seq position) preserves
        events.sort(key=lambda ev: (ev[0] + ev[2], ev[0]))
        for p, kind, q, ln in events:
            if kind == 0:
                out.append(VcfRecord(contig, p + 1, (ref[p], seq[q]), 60.0))
            elif kind == 1:
                if p > 0:
                    out.append(VcfRecord(contig, p, (ref[p - 1:p + ln], ref[p - 1]), 60.0))
            else:
                if p > 0 and q > 0:
                    out.append(VcfRecord(contig, p, (ref[p - 1], ref[p - 1] + seq[q:q + ln]), 60.0))
    return out


def merge_records(recs1, recs2, regions):
    """Two haplotype record lists -> diploid records with GT (src/vcf.py:141-210): same position and
    same alleles -> 1|1; same position, different alleles -> two lines 1|0 and 0|1; else 1|0 / 0|1."""
    out = []
    for contig, start, stop in regions:
        a, b = fetch(recs1, contig, start, stop), fetch(recs2, contig, start, stop)
        i = j = 0
        while i < len(a) or j < len(b):
            p1 = a[i].pos if i < len(a) else float("inf")
            p2 = b[j].pos if j < len(b) else float("inf")
            pos = min(p1, p2)
            h1, h2 = p1 == pos, p2 == pos
            if h1 and h2:
                if a[i].alleles == b[j].alleles:
                    out.append(_with_gt(a[i], (1, 1)))
                else:
                    out.append(_with_gt(a[i], (1, 0)))
                    out.append(_with_gt(b[j], (0, 1)))
            elif h1:
                out.append(_with_gt(a[i], (1, 0)))
            else:
                out.append(_with_gt(b[j], (0, 1)))
            i += h1
            j += h2
    return out


def _with_gt(rec, gt):
    r = _hap_record(rec)
    r.gt = gt
    return r


# ---------------------------------------------------------------------------
# writing
GEN_HEADER_TAIL = ['##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
                   '##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="Genotype quality score">']


def gen_header(hap_data):
    """Header of a generated VCF (src/vcf.py:279-291, 403-408): the contig lines carry the lengths of
    the standardised contigs, inserted after the first two lines."""
    return (["##fileformat=VCFv4.2", '##FILTER=<ID=PASS,Description="All filters passed">'] +
            [f"##contig=<ID={c},length={len(ref)}>" for c, _h, _s, ref, _c in hap_data] + GEN_HEADER_TAIL)


def _fmt_qual(q):
    if q is None:
        return "."
    return str(int(q)) if float(q).is_integer() else f"{q:g}"


def format_record(rec, sample_cols=1, sep="|"):
    """One VCF line.  GT is written phased ('|'): the two columns ARE the two haplotypes the records came
    from.  (pysam leaves a freshly set GT unphased, so the reference's own output shows '/'.)"""
    alts = ",".join(rec.alleles[1:]) if len(rec.alleles) > 1 else "."
    f = [rec.contig, str(rec.pos), rec.id, rec.alleles[0], alts, _fmt_qual(rec.qual), rec.filter, rec.info]
    if sample_cols:
        if rec.gt:
            f += ["GT"] + [sep.join("." if g is None else str(g) for g in rec.gt)] * sample_cols
        else:
            f += ["GT"] + ["."] * sample_cols
    return "\t".join(f)


def bgzf_write(path, data, level=6):
    """BGZF (blocked gzip, what bgzip/tabix and htslib read) of a byte string."""
    with open(path, "wb") as fh:
        for p in range(0, len(data), 0xFF00):
            chunk = data[p:p + 0xFF00]
            comp = zlib.compressobj(level, zlib.DEFLATED, -15)
            body = comp.compress(chunk) + comp.flush()
            fh.write(struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 0xFF, 6, 66, 67, 2, len(body) + 25))
            fh.write(body)
            fh.write(struct.pack("<II", zlib.crc32(chunk), len(chunk)))
        fh.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))


def write_vcf(path, header, records, samples=("SAMPLE",)):
    """Write header + records as BGZF (`.gz`) or plain text; index with tabix when there is one."""
    cols = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO" + ("\tFORMAT\t" + "\t".join(samples) if samples else "")
    text = "\n".join(list(header) + [cols] + [format_record(r, len(samples)) for r in records]) + "\n"
    if path.endswith(".gz"):
        bgzf_write(path, text.encode())
        if shutil.which("tabix"):
            subprocess.run(["tabix", "-f", "-p", "vcf", path])
    else:
        with open(path, "w") as fh:
            fh.write(text)
    return path
