"""BAM in / SAM out around align(): counterparts of reference src/bam.pyx:18-89,
127-145 (get_read_data, realign_read, create_header) and src/util.py:16-93
(get_bam_regions), with a batch in the middle instead of one align() per call.

pysam is not needed: BAM is BGZF (concatenated gzip members) around a simple
binary record stream; the reference bases come from the FASTA slice
[reference_start, reference_start + reference_length), which is what pysam's
get_reference_sequence() reconstructs from the MD tag.

Two implementations of the same logic live here:
  * NativeBam / NativeFasta / realign_native: thin ctypes wrappers of the library's
    C++ host I/O (csrc/hostio.hpp: parallel BGZF inflate, batch packing, SAM
    formatting) -- what realign.py runs;
  * BamFile / read_fasta / get_read_data / realign_reads: the pure-Python
    restatement, record by record as the reference does it -- what the tests
    compare the native path with.
write_bam() makes small BAM files for tests and benchmarks.
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

    def refs_with_reads(self):
        return {r.ref_id for r in self.records if r.ref_id >= 0}

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
        with_reads = bam.refs_with_reads()
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
            fh.write(sam_line(rd, final))
    return len(read_data)


def realign_haps(ctx, hap_data, r=30, max_b_rows=20000):
    """Batched realign_hap (src/bam.pyx:93-123), the second caller of align() in the reference
    (standardize_vcf.py:32-34: whole haplotype sequences against the reference, thousands of
    independent chunks per sequence): [(contig, hap, seq, ref, cigar)] -> the same tuples with the
    realigned, standardised, expanded 'MID' CIGAR."""
    hap_data = list(hap_data)
    if not hap_data:
        return []
    refs = [bases_to_int(h[3]) for h in hap_data]
    seqs = [bases_to_int(h[2]) for h in hap_data]
    alns, status = ctx.align_batch(refs, seqs, [h[4] for h in hap_data], r=r, max_b_rows=max_b_rows, return_status=True)
    for h, st in zip(hap_data, status):
        if st:
            print(f"\nERROR: inconsistent traceback for {h[0]} hap {h[1]} (status {int(st)})")
    finals = standardize_batch(alns, refs, seqs, expanded=True)
    return [(h[0], h[1], h[2], h[3], f) for h, f in zip(hap_data, finals)]


def sam_line(rd, final):
    """SAM record of one get_read_data tuple with its final CIGAR (src/bam.pyx:83)."""
    read_id, flag, ref_name, start, mapq, _cig, stop, sseq, quals, _ref, hap = rd
    return (f"{read_id}\t{flag}\t{ref_name}\t{start + 1}\t{mapq}\t{final}\t*\t0\t"
            f"{stop - start}\t{sseq}\t{quals}\tHP:i:{hap}\n")


# ---------------------------------------------------------------------------
# native host I/O (libnpore_amd.so, csrc/hostio.hpp)
class NativeFasta:
    """Contig names / lengths of a FASTA held by the library; behaves like {name: sized} for get_bam_regions."""

    class _Sized:
        def __init__(self, n):
            self._n = n

        def __len__(self):
            return self._n

    def __init__(self, path):
        from . import _lib
        self._lib = _lib.load()
        self.handle = self._lib.npore_fasta_open(os.fsencode(path))
        if not self.handle:
            print(f"\nERROR: could not open --ref FASTA '{path}'.")
            sys.exit(1)
        self.names = [self._lib.npore_fasta_name(self.handle, i).decode() for i in range(self._lib.npore_fasta_n(self.handle))]
        self._len = {n: int(self._lib.npore_fasta_len(self.handle, i)) for i, n in enumerate(self.names)}

    def __contains__(self, name):
        return name in self._len

    def __getitem__(self, name):
        return NativeFasta._Sized(self._len[name])

    def __iter__(self):
        return iter(self.names)

    def sequence(self, name):
        """The upper-cased bases of a contig as a str (a copy)."""
        import ctypes as C
        i = len(self.names) - 1 - self.names[::-1].index(name)       # a repeated name: the last one, like a dict
        return C.string_at(self._lib.npore_fasta_seq(self.handle, i), self._len[name]).decode()

    def close(self):
        if self.handle:
            self._lib.npore_fasta_close(self.handle)
            self.handle = None


class NativeFastaSeqs:
    """{contig: upper-cased sequence}, like read_fasta(), backed by the library's parallel parser; a contig's
    str is made on first use (callers that only need a few contigs of a genome)."""

    def __init__(self, path):
        self._fa = NativeFasta(path)
        self._cache = {}

    def __contains__(self, name):
        return name in self._fa

    def __iter__(self):
        return iter(dict.fromkeys(self._fa.names))

    def __len__(self):
        return len(dict.fromkeys(self._fa.names))

    def __getitem__(self, name):
        if name not in self._cache:
            if name not in self._fa:
                raise KeyError(name)
            self._cache[name] = self._fa.sequence(name)
        return self._cache[name]

    def items(self):
        return ((n, self[n]) for n in self)

    def keys(self):
        return list(self)


class OnePassUnsupported(RuntimeError):
    """NativeBam.realign_sequential cannot serve these regions / this file in one pass (take the indexed path)."""


class NativeBam:
    """A BAM file inflated and indexed by the library (same attributes as BamFile where realign needs them)."""

    def __init__(self, path, threads=0, share=None, stream=None, one_pass=False):
        """stream: None = the library decides (files above NPORE_BAM_STREAM_MB, default 1 GB, are STREAMED: no
        inflated copy, 22 bytes of index per record, each batch inflates the blocks its reads lie in), True / False force it.
        share: with several processes per node (one per GPU, torch.distributed.run) local rank 0 does the expensive
        part once and leaves it under /dev/shm for the other local ranks -- the record index of a streamed file
        (npore_bam_save_index), the inflated stream of a small one (npore_bam_dump_inflated); None = do so when
        LOCAL_WORLD_SIZE > 1."""
        from . import _lib
        self._lib = _lib.load()
        self._shared = None
        self.path = path
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if stream is None and os.environ.get("NPORE_BAM_STREAM") in ("0", "1"):
            stream = os.environ["NPORE_BAM_STREAM"] == "1"
        mode = 3 if one_pass else 0 if stream is None else (2 if stream else 1)
        self.one_pass = bool(one_pass)
        if one_pass:
            share = False
        if share is None:
            share = local_world > 1 and os.path.isdir("/dev/shm") and os.environ.get("NPORE_SHARE_BAM", "1") != "0"
        will_stream = mode == 2 or (mode == 0 and self._auto_streams(path))
        self.handle = None
        if share and local_world > 1 and will_stream:
            self.handle = self._open_with_shared_index(path, local_rank, threads)
        elif share and local_world > 1:
            path_to_open = self._shared_copy(path, local_rank, threads)
            self.handle = self._lib.npore_bam_open_mode(os.fsencode(path_to_open), threads, 1, None)
        if not self.handle:
            self.handle = self._lib.npore_bam_open_mode(os.fsencode(path), threads, mode, None)
        if not self.handle:
            msg = _lib.last_error()
            print(f"\nERROR: BAM file '{path}' not found." if "not found" in msg else f"\nERROR: {msg}.")
            sys.exit(1)
        self.streamed = bool(self._lib.npore_bam_is_streamed(self.handle))
        n = self._lib.npore_bam_n_refs(self.handle)
        self.references = [self._lib.npore_bam_ref_name(self.handle, i).decode() for i in range(n)]
        self.lengths = [int(self._lib.npore_bam_ref_len(self.handle, i)) for i in range(n)]
        self.n_records = int(self._lib.npore_bam_n_records(self.handle))

    @staticmethod
    def is_bgzf(path):
        try:
            with open(path, "rb") as fh:
                return fh.read(2) == b"\x1f\x8b"
        except OSError:
            return False

    @staticmethod
    def _auto_streams(path):
        try:
            with open(path, "rb") as fh:
                gz = fh.read(2) == b"\x1f\x8b"
            return gz and os.path.getsize(path) > int(os.environ.get("NPORE_BAM_STREAM_MB", "1024")) * (1 << 20)
        except OSError:
            return False

    _opens = {}          # (path, local rank) -> how many times this process has opened it for sharing (the same in every local rank)

    @classmethod
    def _shm_key(cls, path):
        """Name of the files local rank 0 leaves for the other local ranks.  Besides the file's identity it holds only
        what the LAUNCHER hands to every local rank alike -- the rendezvous (MASTER_ADDR : MASTER_PORT), torchrun's run
        id, a batch system's job id -- and the count of this process's opens of the path (SPMD code opens the same
        files in the same order on every rank; a second NativeBam of the same file in one run then does not find the
        first one's `.skip`).  Nothing per-process goes in (rounds 3 - 4 hashed the parent's pid: local ranks started
        by per-rank wrapper scripts have different parents and never met).  Two launches that share all of this share
        the key; local rank 0 removes what an earlier one left (`_announce_maker`), and a rank whose key does diverge
        for a reason not foreseen here gives up after a grace period (`_wait_for_maker`)."""
        import hashlib
        st = os.stat(path)
        ap = os.path.abspath(path)
        who = (ap, os.environ.get("LOCAL_RANK", "0"))
        gen = cls._opens[who] = cls._opens.get(who, 0) + 1
        job = ":".join(os.environ.get(k, "") for k in ("MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "SLURM_JOB_ID",
                                                         "SLURM_STEP_ID", "PBS_JOBID", "LSB_JOBID", "NPORE_JOB_ID"))
        ident = f"{ap}:{st.st_size}:{st.st_mtime_ns}:{job}:{gen}"
        return hashlib.sha1(ident.encode()).hexdigest()[:16]

    @staticmethod
    def _announce_maker(key):
        """local rank 0: its pid where the waiting ranks can see whether it is still alive; stale leftovers of the key removed"""
        for ext in (".idx", ".raw", ".skip"):
            try:
                os.remove(f"/dev/shm/npore_bam_{key}{ext}")
            except OSError:
                pass
        tmp = f"/dev/shm/npore_bam_{key}.pid.tmp"
        with open(tmp, "w") as fh:
            fh.write(str(os.getpid()))
        os.replace(tmp, f"/dev/shm/npore_bam_{key}.pid")

    @staticmethod
    def _wait_for_maker(key, data, skip, timeout):
        """other local ranks: True once `data` is there; False if the maker gave up (`skip`), died, never showed up
        (no pid file within NPORE_SHARE_GRACE_S, 30 s: the ranks' keys differ, or local rank 0 is not running this
        code -- the caller then opens the file itself), or `timeout` passed"""
        import time
        pidfile = f"/dev/shm/npore_bam_{key}.pid"
        grace = float(os.environ.get("NPORE_SHARE_GRACE_S", "30"))
        t0 = time.time()
        seen_maker = False
        while time.time() - t0 < timeout:
            if os.path.exists(data):
                return True
            if os.path.exists(skip):
                return False
            try:
                pid = int(open(pidfile).read())                # (no pid file yet: the maker has not started)
                seen_maker = True
            except (OSError, ValueError):
                pid = None
            if not seen_maker and time.time() - t0 > grace:
                return False
            if pid is not None:
                try:
                    os.kill(pid, 0)
                except ProcessLookupError:                     # the maker is gone: what it left is all there will be
                    return os.path.exists(data)
                except OSError:
                    pass
            time.sleep(0.05)
        return False

    def _open_with_shared_index(self, path, local_rank, threads):
        """Streamed file, several local ranks: local rank 0 makes the record index (one pass over the file) and saves
        it under /dev/shm; the others wait for it and open with it.  Returns a handle or None (caller opens normally)."""
        try:
            key = self._shm_key(path)
        except OSError:
            return None
        ix, skip = f"/dev/shm/npore_bam_{key}.idx", f"/dev/shm/npore_bam_{key}.skip"
        if local_rank == 0:
            self._announce_maker(key)
            h = self._lib.npore_bam_open_mode(os.fsencode(path), threads, 2, None)
            if not h or self._lib.npore_bam_save_index(h, os.fsencode(ix + ".tmp")) != 0:
                open(skip, "w").close()
                self._shared = (ix, skip, key)
                return h or None
            os.replace(ix + ".tmp", ix)
            self._shared = (ix, skip, key)
            return h
        if self._wait_for_maker(key, ix, skip, float(os.environ.get("NPORE_SHARE_WAIT_S", "3600"))):
            return self._lib.npore_bam_open_mode(os.fsencode(path), threads, 2, os.fsencode(ix)) or None
        return None

    def _shared_copy(self, path, local_rank, threads):
        """Path to open: the inflated copy under /dev/shm (local rank 0 makes it, the others wait for it), or
        `path` itself when sharing is not possible (no room, or the maker gave up: a `.skip` marker)."""
        import shutil
        try:
            key = self._shm_key(path)
        except OSError:
            return path
        raw, skip = f"/dev/shm/npore_bam_{key}.raw", f"/dev/shm/npore_bam_{key}.skip"
        if local_rank == 0:
            self._announce_maker(key)
            self._shared = (raw, skip, key)
            h = self._lib.npore_bam_open_mode(os.fsencode(path), threads, 1, None)
            if not h:
                open(skip, "w").close()
                return path
            ok = False
            try:
                size = int(self._lib.npore_bam_inflated_size(h))
                if size * 4 <= shutil.disk_usage("/dev/shm").free:
                    ok = self._lib.npore_bam_dump_inflated(h, os.fsencode(raw + ".tmp")) == 0
            finally:
                self._lib.npore_bam_close(h)
            if not ok:
                open(skip, "w").close()
                return path
            os.replace(raw + ".tmp", raw)
            return raw
        return raw if self._wait_for_maker(key, raw, skip, float(os.environ.get("NPORE_SHARE_WAIT_S", "1800"))) else path

    def refs_with_reads(self):
        return {i for i in range(len(self.references)) if self._lib.npore_bam_ref_has_reads(self.handle, i)}

    def select(self, regions, max_reads=0):
        """Record indices of the reads get_read_data would yield for [(contig, start, stop)]."""
        ids = {n: i for i, n in enumerate(self.references)}
        rid = np.array([ids.get(c, -2) for c, _, _ in regions], np.int32)
        beg = np.array([s for _, s, _ in regions], np.int64)
        end = np.array([e for _, _, e in regions], np.int64)
        # two calls: the count first (cap = 0 writes nothing), then exactly that many entries -- a read overlapping
        # several regions is listed once per region, so n_records * n_regions is the only a-priori bound
        args = (self.handle, len(regions), rid.ctypes.data, beg.ctypes.data, end.ctypes.data, int(max_reads or 0))
        k = self._lib.npore_bam_select(*args, None, 0)
        if k < 0:
            from . import _lib
            raise RuntimeError(_lib.last_error())
        out = np.zeros(max(int(k), 1), np.int64)
        k2 = self._lib.npore_bam_select(*args, out.ctypes.data, len(out))
        assert k2 == k
        return out[:k]

    def fasta_map(self, fasta):
        """int32[n_refs]: index of each BAM reference in the FASTA (-1 if absent)."""
        pos = {n: i for i, n in enumerate(fasta.names)}
        return np.array([pos.get(n, -1) for n in self.references], np.int32)

    def pack(self, fasta, idx, threads=0):
        """(refs, ref_off, seqs, seq_off, cigs, cig_off) for npore_align_batch."""
        idx = np.ascontiguousarray(idx, np.int64)
        n = len(idx)
        ro, so, co = (np.zeros(n + 1, np.int64) for _ in range(3))
        self._check(self._lib.npore_bam_pack_sizes(self.handle, idx.ctypes.data, n, ro.ctypes.data, so.ctypes.data, co.ctypes.data))
        refs = np.zeros(int(ro[-1]) + 64, np.uint8)
        seqs = np.zeros(int(so[-1]) + 64, np.uint8)
        cigs = np.zeros(int(co[-1]) + 64, np.uint8)
        fmap = self.fasta_map(fasta)
        self._check(self._lib.npore_bam_pack(self.handle, fasta.handle, fmap.ctypes.data, idx.ctypes.data, n, refs.ctypes.data,
                                             ro.ctypes.data, seqs.ctypes.data, so.ctypes.data, cigs.ctypes.data,
                                             co.ctypes.data, threads))
        return refs, ro, seqs, so, cigs, co

    def format_sam(self, idx, finals, status, threads=0):
        """SAM text of the selected reads given their final collapsed CIGAR strings."""
        import ctypes as C
        idx = np.ascontiguousarray(idx, np.int64)
        n = len(idx)
        fb = [f.encode() for f in finals]
        fo = np.zeros(n + 1, np.int64)
        np.cumsum([len(f) for f in fb], out=fo[1:])
        fl = np.diff(fo)
        buf = np.frombuffer(b"".join(fb) + b"\0", np.uint8)
        st = np.ascontiguousarray(status, np.int32)
        sam, sam_len = C.c_void_p(), C.c_int64()
        self._check(self._lib.npore_bam_format_sam(self.handle, idx.ctypes.data, n, buf.ctypes.data, fo.ctypes.data,
                                                   fl.ctypes.data, st.ctypes.data, threads, C.byref(sam), C.byref(sam_len)))
        return C.string_at(sam.value, sam_len.value).decode() if sam_len.value else ""

    @staticmethod
    def bai_path(path):
        """the file's .bai index (`x.bam.bai`, or `x.bai` beside `x.bam`), or None"""
        for cand in (path + ".bai", os.path.splitext(path)[0] + ".bai"):
            if os.path.exists(cand):
                return cand
        return None

    def set_share(self, rank, world, bai=None):
        """Several processes on this one-pass handle: process `rank` of `world` will walk a contiguous stretch of the record
        stream, cut at virtual offsets of the .bai linear index (npore_bam_set_share).  Raises OnePassUnsupported when
        there is no usable index (every rank then takes the indexed reader: the decision depends on the files alone)."""
        import ctypes as C
        bai = bai or self.bai_path(self.path)
        rc = self._lib.npore_bam_set_share(self.handle, int(rank), int(world), os.fsencode(bai) if bai else None)
        if rc == -5:
            from . import _lib
            raise OnePassUnsupported(_lib.last_error())
        self._check(rc)
        info = np.zeros(4, np.int64)
        self._check(self._lib.npore_bam_share_info(self.handle, info.ctypes.data))
        return tuple(int(x) for x in info)

    def realign_sequential(self, ctx, fasta, regions, out_path, batch_reads=4000, max_reads=0, r=30, max_b_rows=20000,
                           indel_start=5.0, indel_extend=1.0, threads=0, bad_cap=1000):
        """ONE PASS over the file: inflate, filter by `regions` [(contig, start, stop)] (at most one per contig, in header
        order), batch, realign, write -- npore_bam_realign_sequential.  Returns (reads selected, [(ordinal, status)] of
        the first bad reads, (refused, inconsistent)); raises OnePassUnsupported when the regions or the file's
        order rule the one-pass run out (the caller truncates the output and takes the indexed path)."""
        ids = {n: i for i, n in enumerate(self.references)}
        rid = np.array([ids.get(c, -2) for c, _, _ in regions], np.int32)
        beg = np.array([s for _, s, _ in regions], np.int64)
        end = np.array([e for _, _, e in regions], np.int64)
        if len(rid) and ((rid < 0).any() or (np.diff(rid) <= 0).any()):
            raise OnePassUnsupported("regions are not one per contig in header order")
        counts = np.zeros(3, np.int64)
        bad_ord, bad_st = np.zeros(max(bad_cap, 1), np.int64), np.zeros(max(bad_cap, 1), np.int32)
        fmap = self.fasta_map(fasta)
        rc = self._lib.npore_bam_realign_sequential(ctx.handle, self.handle, fasta.handle, fmap.ctypes.data, len(regions), rid.ctypes.data,
                                                    beg.ctypes.data, end.ctypes.data, int(max_reads or 0), int(batch_reads), indel_start,
                                                    indel_extend, max_b_rows, r, threads, os.fsencode(out_path), counts.ctypes.data,
                                                    bad_ord.ctypes.data, bad_st.ctypes.data, bad_cap)
        if rc == -5:
            from . import _lib
            msg = _lib.last_error()
            # only what rules the ONE-PASS run out sends the caller to the indexed reader; a band or chunk height the
            # kernels do not cover (the same code) would fail there in the same way, after a whole indexing pass
            if "one-pass ingest" in msg or "not sorted by reference" in msg:
                raise OnePassUnsupported(msg)
        self._check(rc)
        nb = int(min(bad_cap, counts[1] + counts[2]))
        return int(counts[0]), list(zip(bad_ord[:nb].tolist(), bad_st[:nb].tolist())), (int(counts[1]), int(counts[2]))

    def realign_batch(self, ctx, fasta, idx, r=30, max_b_rows=20000, indel_start=5.0, indel_extend=1.0, threads=0):
        """(SAM text, status[n]) of one batch: pack -> GPU align -> standardise -> format, all in the library."""
        import ctypes as C
        idx = np.ascontiguousarray(idx, np.int64)
        n = len(idx)
        st = np.zeros(max(n, 1), np.int32)
        fmap = self.fasta_map(fasta)
        sam, sam_len = C.c_void_p(), C.c_int64()
        self._check(self._lib.npore_bam_realign_batch(ctx.handle, self.handle, fasta.handle, fmap.ctypes.data, idx.ctypes.data, n,
                                                      indel_start, indel_extend, max_b_rows, r, threads,
                                                      C.byref(sam), C.byref(sam_len), st.ctypes.data))
        # a view of the library's buffer (valid until the next call on this handle): no copy before the file write
        return (memoryview((C.c_char * sam_len.value).from_address(sam.value)) if sam_len.value else memoryview(b"")), st[:n]

    def realign_file(self, ctx, fasta, idx, out_sam, batch_reads=4000, r=30, max_b_rows=20000, indel_start=5.0,
                     indel_extend=1.0, threads=0):
        """All selected reads, batch by batch, appended to out_sam by the library with packing, GPU work and
        formatting/writing of neighbouring batches overlapped.  Returns status[n]."""
        idx = np.ascontiguousarray(idx, np.int64)
        st = np.zeros(max(len(idx), 1), np.int32)
        fmap = self.fasta_map(fasta)
        self._check(self._lib.npore_bam_realign_file(ctx.handle, self.handle, fasta.handle, fmap.ctypes.data, idx.ctypes.data,
                                                     len(idx), int(batch_reads), indel_start, indel_extend, max_b_rows, r,
                                                     threads, os.fsencode(out_sam), st.ctypes.data))
        return st[:len(idx)]

    def timing(self):
        """Host wall time (ms) of the stages of the last realign_batch."""
        ms = np.zeros(4, np.float64)
        self._lib.npore_bam_last_timing(self.handle, ms.ctypes.data, 4)
        return dict(zip(("pack_ms", "align_ms", "standardize_ms", "format_ms"), ms.tolist()))

    def file_timing(self):
        """Stage clocks (ms) of the last realign_file: per-stage sums over the batches (stages overlap), the wall time
        of the call, the GPU's kernel and PCIe time (npore_bam_file_timing)."""
        ms = np.zeros(8, np.float64)
        self._lib.npore_bam_file_timing(self.handle, ms.ctypes.data, 8)
        return dict(zip(("fetch_pack_ms", "align_call_ms", "standardize_ms", "format_ms", "write_ms", "wall_ms", "gpu_kernels_ms",
                         "pcie_ms"), ms.tolist()))

    def _check(self, rc):
        if rc != 0:
            from . import _lib
            raise RuntimeError(f"libnpore_amd: {rc} {_lib.last_error()}")

    def close(self):
        if self.handle:
            self._lib.npore_bam_close(self.handle)
            self.handle = None
        if self._shared:                 # the maker takes the shared copy away; a rank that comes later inflates itself
            raw, skip, key = self._shared
            self._shared = None
            try:
                open(skip, "w").close()
            except OSError:
                pass
            for f in (raw, raw + ".tmp", f"/dev/shm/npore_bam_{key}.pid"):
                try:
                    os.remove(f)
                except OSError:
                    pass
            import atexit                # the marker for late comers goes when this process does
            atexit.register(lambda f=skip: os.path.exists(f) and os.remove(f))


def realign_native(ctx, bam, fasta, idx, out_sam, r=30, max_b_rows=20000, batch_reads=0, threads=0):
    """realign_reads() through the library; returns the number of reads handed in.  batch_reads > 0: the whole
    index list in overlapped batches written by the library itself; 0: one batch, text written here.
    threads: host threads of the parallel host stages (0 = all cores; one process per GPU: dist.host_threads_per_rank)."""
    if len(idx) == 0:
        return 0
    if batch_reads > 0:
        text, status = None, bam.realign_file(ctx, fasta, idx, out_sam, batch_reads=batch_reads, r=r, max_b_rows=max_b_rows,
                                              threads=threads)
    else:
        text, status = bam.realign_batch(ctx, fasta, idx, r=r, max_b_rows=max_b_rows, threads=threads)
    bad = np.nonzero(status)[0]
    for k in bad:
        if status[k] & 32:
            print(f"\nERROR: read #{int(idx[k])}: CIGAR does not match sequence lengths; skipped.")
        else:
            print(f"\nERROR: inconsistent traceback for read #{int(idx[k])} (status {int(status[k])})")   # src/aln.pyx:689-716
    if text is not None:
        with open(out_sam, "ab") as fh:
            fh.write(text)
    return len(idx)


# ---- confusion matrices (reference src/bam.pyx:166-200, 301-316, 351-499) ----------------------------------------
def get_pileups(bam_path, ctg, start, end):
    """Column 5 of `samtools mpileup -r ctg:start+1-end bam`, upper-cased, one string per reported position
    (reference src/bam.pyx:301-316).  Needs samtools on PATH, like the reference."""
    import shutil
    import subprocess
    if not shutil.which("samtools"):
        print("\nERROR: recalculating the confusion matrices needs `samtools` (mpileup) on PATH.")
        sys.exit(1)
    pile = subprocess.Popen(["samtools", "mpileup", "-r", f"{ctg}:{start + 1}-{end}", bam_path],
                            stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    for line in pile.stdout:
        f = line.decode("utf-8").rstrip("\n").split("\t")
        yield (f[4] if len(f) > 4 else "").upper().strip()


def calc_confusion_matrices(range_tuple, pileups=None, refs=None, np_info=None, threads=0):
    """Reference src/bam.pyx:351-499 for one range (ctg, start, end): (subs[5,5], nps[max_n,max_l+1,max_l+1],
    inss[max_l+1], dels[max_l+1]) int64 counts of the basecaller's errors against the reference.
    pileups: iterable of column-5 strings (default: samtools, get_pileups); refs: {contig: sequence} (default
    cfg.args.refs); np_info: get_np_info of refs[ctg][start:end+1] (default: aln.get_np_info, on the GPU).
    The character loop runs in the library (csrc/confusion.hpp) on all host cores."""
    import ctypes as C
    from . import _lib, aln
    from .cig import bases_to_int
    lib = _lib.load()
    ctg, start, end = range_tuple
    refs = cfg.args.refs if refs is None else refs
    max_n, max_l = int(cfg.args.max_n), int(cfg.args.max_l)
    contig = refs[ctg]
    if pileups is None:
        pileups = get_pileups(cfg.args.bam, ctg, start, end)
    lines = [p.upper().strip().encode() for p in pileups]
    if np_info is None:
        np_info = aln.get_np_info(bases_to_int(contig[start:end + 1]))
    np_info = np.ascontiguousarray(np_info, dtype=np.int32)
    codes = np.ascontiguousarray(bases_to_int(contig[start:end]), dtype=np.uint8)
    # (the character loop only looks at ref_text[pos+1 .. pos+1+max_n) of the range's positions; Python's slice clipping
    # at the contig end is what the reference's contig[...] slices do there)
    text = contig[start:end + max_n + 1].upper().encode()
    off = np.zeros(len(lines) + 1, np.int64)
    np.cumsum([len(x) for x in lines], out=off[1:])
    buf = b"".join(lines) + b"\0"
    subs = np.zeros((cfg.nbases, cfg.nbases), np.int64)
    nps = np.zeros((max_n, max_l + 1, max_l + 1), np.int64)
    inss = np.zeros(max_l + 1, np.int64)
    dels = np.zeros(max_l + 1, np.int64)
    bad = C.c_int64(0)
    rc = lib.npore_confusion_counts(buf, off.ctypes.data, len(lines), codes.ctypes.data, len(codes), text, len(text),
                                    np_info.ctypes.data, len(np_info), max_n, max_l, subs.ctypes.data, nps.ctypes.data,
                                    inss.ctypes.data, dels.ctypes.data, C.byref(bad), threads)
    if rc:
        raise RuntimeError(_lib.last_error())
    if bad.value:
        print(f"ERROR: unexpected character in {bad.value} pileup line(s) of {ctg}:{start}-{end}.")   # src/bam.pyx:473-476
    return subs, nps, inss, dels


def get_confusion_matrices():
    """Reference src/bam.pyx:166-200: the cached count matrices of --stats_dir, or (--recalc_cms) counted from the
    BAM range by range (cfg.args.regions cut into --chunk_width pieces), summed and cached there.
    Loading defaults to the shipped guppy5_stats; a recount is written to --stats_dir (default ./stats, like the
    reference) and NEVER into the package's data directory.  Several processes (torch.distributed.run): rank 0
    recounts and writes (temp file + rename), the others wait at a barrier and load."""
    shipped = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "guppy5_stats")
    names = ("subs", "nps", "inss", "dels")
    if not getattr(cfg.args, "recalc_cms", False):
        d = cfg.args.stats_dir or shipped
        print("> loading confusion matrices")
        return tuple(np.load(os.path.join(d, f"{k}_cm.npy")) for k in names)
    d = cfg.args.stats_dir or "./stats"
    if os.path.realpath(d) == os.path.realpath(shipped):
        print("\nERROR: --recalc_cms would overwrite the shipped guppy5_stats tables; give another --stats_dir.")
        sys.exit(1)
    from . import dist as dist_mod

    def recount():
        print("> calculating confusion matrices")
        from .bed import get_ranges
        total = None
        ranges = get_ranges(cfg.args.regions, cfg.args.chunk_width)
        for k, rg in enumerate(ranges):
            res = calc_confusion_matrices(rg)
            total = res if total is None else tuple(a + b for a, b in zip(total, res))
            print(f"\r    {k + 1} of {len(ranges)} chunks processed.", end="", flush=True)
        print(" ")
        if total is None:
            total = calc_confusion_matrices(("", 0, 0), pileups=[], refs={"": ""},
                                            np_info=np.zeros((0, 2, int(cfg.args.max_n)), np.int32))
        os.makedirs(d, exist_ok=True)
        for k, m in zip(names, total):
            tmp = os.path.join(d, f".{k}_cm.{os.getpid()}.tmp.npy")
            np.save(tmp, m)
            os.replace(tmp, os.path.join(d, f"{k}_cm.npy"))
        return total

    # several ranks: rank 0 recounts; the others learn whether it succeeded (no barrier that never comes, no 30-minute
    # default timeout on a step that takes hours on a genome)
    total = dist_mod.rank0_then_all(recount)
    if total is None:
        total = tuple(np.load(os.path.join(d, f"{k}_cm.npy")) for k in names)
    if getattr(cfg.args, "recalc_exit", False):
        sys.exit(0)
    return total


def write_bai(bam_path, bai_path=None):
    """A .bai for a BAM file (tests / benchmarks; `samtools index` makes the real ones): the LINEAR index only -- per
    reference and 16 kb window the virtual offset (block offset << 16 | offset in the block) of the first record that
    overlaps the window, SAM specification 5.2 -- with no bins, which is all npore_bam_set_share reads."""
    raw = open(bam_path, "rb").read()
    blocks, p, u = [], 0, 0                      # (compressed offset, inflated offset) of every BGZF block
    parts = []
    while p < len(raw):
        xlen = struct.unpack_from("<H", raw, p + 10)[0]
        bsize = struct.unpack_from("<H", raw, p + 16)[0] + 1
        data = zlib.decompress(raw[p + 12 + xlen:p + bsize - 8], -15)
        blocks.append((p, u))
        parts.append(data)
        u += len(data)
        p += bsize
    data = b"".join(parts)
    starts = np.array([b[1] for b in blocks], np.int64)
    coffs = np.array([b[0] for b in blocks], np.int64)
    l_text, = struct.unpack_from("<i", data, 4)
    q = 8 + l_text
    n_ref, = struct.unpack_from("<i", data, q); q += 4
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", data, q); q += 8 + l_name
    lin = [dict() for _ in range(n_ref)]
    ref_len_of_op = (1, 0, 1, 1, 0, 0, 0, 1, 1)  # MIDNSHP=X consume the reference?
    while q + 4 <= len(data):
        bs, = struct.unpack_from("<i", data, q)
        rid, pos, l_rn, _mq, _bin, n_cig, _flag, _l_seq = struct.unpack_from("<iiBBHHHi", data, q + 4)
        k = int(np.searchsorted(starts, q, side="right")) - 1
        while starts[k] == q and k > 0 and starts[k - 1] == q:      # (empty blocks: the first of them)
            k -= 1
        v = (int(coffs[k]) << 16) | (q - int(starts[k]))
        if rid >= 0:
            cig = struct.unpack_from(f"<{n_cig}I", data, q + 36 + l_rn)
            end = pos + max(1, sum((c >> 4) * ref_len_of_op[c & 15] for c in cig))
            for w in range(pos >> 14, ((end - 1) >> 14) + 1):
                lin[rid].setdefault(w, v)
        q += 4 + bs
    out = bytearray(b"BAI\1" + struct.pack("<i", n_ref))
    for d in lin:
        out += struct.pack("<i", 0)                                  # no bins
        n_intv = (max(d) + 1) if d else 0
        out += struct.pack("<i", n_intv)
        last = 0
        for w in range(n_intv):                                      # (windows without a record: the previous entry, as samtools does)
            last = d.get(w, last)
            out += struct.pack("<Q", last)
    bai_path = bai_path or bam_path + ".bai"
    with open(bai_path, "wb") as fh:
        fh.write(bytes(out))
    return bai_path


def write_bam(path, references, records, level=6):
    """Write a BAM file (tests / benchmarks).  references: [(name, length)]; records: dicts with
    name, flag, ref_id, pos, mapq, cigar [(op, len)], seq (str over =ACMGRSVTWYHKDBN), qual (bytes or None),
    hp (int or None)."""
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in references)
    out = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(references)))
    for n, l in references:
        out += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
    code = {c: i for i, c in enumerate(_SEQ16)}
    for r in records:
        name = r["name"].encode() + b"\0"
        seq = r["seq"]
        nib = np.array([code[c] for c in seq] + ([0] if len(seq) & 1 else []), np.uint8)
        packed = ((nib[0::2] << 4) | nib[1::2]).astype(np.uint8).tobytes() if len(seq) else b""
        qual = r.get("qual")
        qual = bytes([0xFF]) * len(seq) if qual is None else bytes(qual)
        cig = b"".join(struct.pack("<I", (ln << 4) | op) for op, ln in r["cigar"])
        aux = b"" if r.get("hp") is None else b"HPC" + bytes([r["hp"]])
        body = struct.pack("<iiBBHHHiiii", r["ref_id"], r["pos"], len(name), r.get("mapq", 60), 4680, len(r["cigar"]),
                           r["flag"], len(seq), -1, -1, 0) + name + cig + packed + qual + aux
        out += struct.pack("<i", len(body)) + body
    with open(path, "wb") as fh:
        for p in range(0, len(out), 0xFF00):      # BGZF blocks of < 64 KiB
            chunk = bytes(out[p:p + 0xFF00])
            comp = zlib.compressobj(level, zlib.DEFLATED, -15)
            data = comp.compress(chunk) + comp.flush()
            fh.write(struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 0xFF, 6, 66, 67, 2, len(data) + 25))
            fh.write(data)
            fh.write(struct.pack("<II", zlib.crc32(chunk), len(chunk)))
        fh.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))   # BGZF EOF block
