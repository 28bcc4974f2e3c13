"""BAM in / SAM out around align(): counterparts of reference src/bam.pyx:18-89,
127-145 (get_read_data, realign_read, create_header) and src/util.py:16-93
(get_bam_regions), with a batch in the middle instead of one align() per call.

pysam is optional: BAM is BGZF (concatenated gzip members) around a simple
binary record stream, which zlib + struct decode directly; the reference bases
come from the FASTA slice [reference_start, reference_start + reference_length),
which is what pysam's get_reference_sequence() reconstructs from the MD tag.
"""
import os
import struct
import sys
import zlib

import numpy as np

from . import cfg
from .cig import bases_to_int, expand_cigar, standardize_batch

_SEQ16 = "=ACMGRSVTWYHKDBN"
_CIGOPS = "MIDNSHP=XB"


def read_fasta(path):
    """{contig: sequence} (upper-cased), contigs in file order."""
    try:
        fh = open(path)
    except (IOError, OSError):
        print(f"\nERROR: could not open --ref FASTA '{path}'.")
        sys.exit(1)
    seqs, name, parts = {}, None, []
    with fh:
        for line in fh:
            if line.startswith(">"):
                if name is not None:
                    seqs[name] = "".join(parts).upper()
                name, parts = line[1:].split()[0], []
            else:
                parts.append(line.strip())
    if name is not None:
        seqs[name] = "".join(parts).upper()
    return seqs


def _bgzf_decompress(path):
    try:
        raw = open(path, "rb").read()
    except FileNotFoundError:
        print(f"\nERROR: BAM file '{path}' not found.")      # reference src/bam.pyx:22-24
        sys.exit(1)
    out, pos = [], 0
    while pos < len(raw):
        d = zlib.decompressobj(31)
        out.append(d.decompress(raw[pos:]))
        used = len(raw) - pos - len(d.unused_data)
        if used <= 0:
            break
        pos += used
    return b"".join(out)


class BamRecord:
    __slots__ = ("query_name", "flag", "ref_id", "reference_start", "mapping_quality", "cigar", "seq", "qual", "hp")


class BamFile:
    """Minimal reader: header text, reference names/lengths, all records."""

    def __init__(self, path):
        data = _bgzf_decompress(path)
        if data[:4] != b"BAM\1":
            print(f"\nERROR: '{path}' is not a BAM file.")
            sys.exit(1)
        l_text, = struct.unpack_from("<i", data, 4)
        self.text = data[8:8 + l_text].decode(errors="replace").rstrip("\0")
        p = 8 + l_text
        n_ref, = struct.unpack_from("<i", data, p)
        p += 4
        self.references, self.lengths = [], []
        for _ in range(n_ref):
            l_name, = struct.unpack_from("<i", data, p)
            name = data[p + 4:p + 4 + l_name - 1].decode()
            l_ref, = struct.unpack_from("<i", data, p + 4 + l_name)
            self.references.append(name)
            self.lengths.append(l_ref)
            p += 8 + l_name
        self.records = []
        while p + 4 <= len(data):
            block_size, = struct.unpack_from("<i", data, p)
            q = p + 4
            ref_id, pos, l_rn, mapq, _bin, n_cig, flag, l_seq, _nref, _npos, _tlen = struct.unpack_from("<iiBBHHHiiii", data, q)
            q += 32
            r = BamRecord()
            r.query_name = data[q:q + l_rn - 1].decode()
            q += l_rn
            cig = struct.unpack_from(f"<{n_cig}I", data, q)
            q += 4 * n_cig
            r.cigar = [(c & 15, c >> 4) for c in cig]
            sb = np.frombuffer(data, np.uint8, (l_seq + 1) // 2, q)
            q += (l_seq + 1) // 2
            nib = np.empty(2 * len(sb), np.uint8)
            nib[0::2] = sb >> 4
            nib[1::2] = sb & 15
            r.seq = "".join(_SEQ16[x] for x in nib[:l_seq])
            r.qual = data[q:q + l_seq]
            q += l_seq
            r.hp = None
            end = p + 4 + block_size
            while q + 3 <= end:      # optional fields: find HP
                tag, typ = data[q:q + 2], chr(data[q + 2])
                q += 3
                if typ in "cCsSiI":
                    fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}[typ]
                    val, = struct.unpack_from(fmt, data, q)
                    q += struct.calcsize(fmt)
                    if tag == b"HP":
                        r.hp = int(val)
                elif typ == "A":
                    q += 1
                elif typ == "f":
                    q += 4
                elif typ in "ZH":
                    z = data.index(b"\0", q)
                    q = z + 1
                elif typ == "B":
                    sub = chr(data[q]); cnt, = struct.unpack_from("<i", data, q + 1)
                    q += 5 + cnt * {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[sub]
                else:
                    break
            r.flag, r.ref_id, r.reference_start, r.mapping_quality = flag, ref_id, pos, mapq
            self.records.append(r)
            p = end


def get_bam_regions(bam, ref_seqs):
    """cfg.args.regions = [(contig, start, end)], reference src/util.py:16-93."""
    a = cfg.args
    if getattr(a, "contig", None):
        if a.contig not in ref_seqs:
            print(f"ERROR: contig '{a.contig}' not present in '{a.ref}'. Valid contigs are: {list(ref_seqs)}")
            sys.exit(1)
        if getattr(a, "contigs", None):
            print("\nERROR: can't set 'contig' and 'contigs'.")
            sys.exit(1)
        beg = a.contig_beg or 0
        max_end = len(ref_seqs[a.contig]) - 1
        end = a.contig_end or max_end
        a.regions = [(a.contig, beg, min(max_end, end))]
    elif getattr(a, "contigs", None):
        if a.contig_beg or a.contig_end:
            print("\nERROR: can't set start/endpoints with multiple contigs.")
            sys.exit(1)
        a.regions = []
        for ctg in a.contigs.split(","):
            if ctg not in ref_seqs:
                print(f"ERROR: contig '{ctg}' not present in '{a.ref}'. Valid contigs are: {list(ref_seqs)}")
                sys.exit(1)
            a.regions.append((ctg, 0, len(ref_seqs[ctg]) - 1))
    elif getattr(a, "bed", None):
        try:
            a.regions = [(c, int(s), int(e)) for c, s, e in (x.strip().split()[:3] for x in open(a.bed) if x.strip())]
        except FileNotFoundError:
            print("\nERROR: could not open 'cfg.args.bed' BED.")
            sys.exit(1)
    else:
        if getattr(a, "contig_beg", None) or getattr(a, "contig_end", None):
            print("\nERROR: 'contig' not supplied, but start/endpoints set.")
            sys.exit(1)
        a.regions = []
        with_reads = {r.ref_id for r in bam.records if r.ref_id >= 0}
        for k, (ctg, l) in enumerate(zip(bam.references, bam.lengths)):
            if ctg not in ref_seqs:
                print(f"WARNING: contig '{ctg}' present in '{a.bam}', but not '{a.ref}', skipping...")
            elif k in with_reads:
                a.regions.append((ctg, 0, l - 1))
    return a.regions


def get_read_data(bam, ref_seqs):
    """Generator of the reference's 11-tuples (src/bam.pyx:18-47): primary mapped reads
    overlapping cfg.args.regions, soft clips trimmed off sequence and qualities."""
    kept = 0
    name_to_id = {n: i for i, n in enumerate(bam.references)}
    for ctg, start, stop in cfg.args.regions:
        rid = name_to_id.get(ctg, -2)
        for r in bam.records:
            if r.ref_id != rid:
                continue
            ref_len = sum(n for op, n in r.cigar if op in (0, 2, 3, 7, 8))
            if not (r.reference_start < stop and r.reference_start + ref_len > start):
                continue
            if cfg.args.max_reads and kept >= cfg.args.max_reads:
                return
            if r.flag & (0x100 | 0x800 | 0x4):          # secondary, supplementary, unmapped
                continue
            kept += 1
            lead = r.cigar[0][1] if r.cigar and r.cigar[0][0] == 4 else 0
            if len(r.cigar) > 1 and r.cigar[0][0] == 5 and r.cigar[1][0] == 4:
                lead = r.cigar[1][1]
            trail = r.cigar[-1][1] if len(r.cigar) > 1 and r.cigar[-1][0] == 4 else 0
            if len(r.cigar) > 2 and r.cigar[-1][0] == 5 and r.cigar[-2][0] == 4:
                trail = r.cigar[-2][1]
            qend = len(r.seq) - trail
            quals = "*" if (not r.qual or r.qual[0] == 0xFF) else "".join(chr(33 + x) for x in r.qual[lead:qend])
            yield (r.query_name, r.flag, ctg, r.reference_start, r.mapping_quality,
                   "".join(f"{n}{_CIGOPS[op]}" for op, n in r.cigar), r.reference_start + ref_len,
                   r.seq[lead:qend].upper(), quals,
                   ref_seqs[ctg][r.reference_start:r.reference_start + ref_len].upper(),
                   0 if r.hp is None else int(r.hp))


def create_header(outfile, bam):
    """SAM header as the reference writes it through pysam (src/bam.pyx:127-145); truncates."""
    if os.path.dirname(outfile):
        os.makedirs(os.path.dirname(outfile), exist_ok=True)
    with open(outfile, "w") as fh:
        fh.write("@HD\tVN:1.6\tSO:coordinate\n")
        for name, l in zip(bam.references, bam.lengths):
            fh.write(f"@SQ\tSN:{name}\tLN:{l}\n")
        fh.write(f"@PG\tPN:realigner\tID:realigner\tVN:{cfg.__version__}\tCL:{' '.join(sys.argv)}\n")


def realign_reads(ctx, read_data, out_sam, r=30, max_b_rows=20000):
    """Batched realign_read (src/bam.pyx:51-84): align on the GPU, standardise, append SAM lines.
    Returns the number of reads written."""
    read_data = list(read_data)
    if not read_data:
        return 0
    cigs, refs, seqs = [], [], []
    for rd in read_data:
        cigs.append(expand_cigar(rd[5]).replace("S", "").replace("H", ""))     # src/bam.pyx:59
        refs.append(bases_to_int(rd[9]))
        seqs.append(bases_to_int(rd[7]))
    alns, status = ctx.align_batch(refs, seqs, cigs, r=r, max_b_rows=max_b_rows, return_status=True)
    finals = standardize_batch(alns, refs, seqs)          # src/bam.pyx:65-78, C++ glue in the library
    with open(out_sam, "a") as fh:
        for rd, final, st in zip(read_data, finals, status):
            read_id, flag, ref_name, start, mapq, _cig, stop, sseq, quals, _ref, hap = rd
            if st & 32:
                print(f"\nERROR: read '{read_id}': CIGAR does not match sequence lengths; skipped.")
                continue
            if st:
                print(f"\nERROR: inconsistent traceback for read '{read_id}' (status {int(st)})")   # src/aln.pyx:689-716
            fh.write(f"{read_id}\t{flag}\t{ref_name}\t{start + 1}\t{mapq}\t{final}\t*\t0\t"
                     f"{stop - start}\t{sseq}\t{quals}\tHP:i:{hap}\n")
    return len(read_data)
