"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the
golden fixtures and against the oracle on seeded inputs.  Bit-exact strings."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import oracle
from npore_amd import _lib, aln, synth
from conftest import load_json, enc, expand_cigar, GOLDEN

pytestmark = pytest.mark.gpu


def sha(s):
    return hashlib.sha256(s.encode()).hexdigest()


_FULLSIZE = None


def check_fullsize_digests(name, got, first=0):
    """Every string of a full-size configuration against the per-read digests the pinned oracle produced in the build
    container (tests/golden/make_fullsize_digests.py -> fullsize_digests.npz: read index, length, sha256[:16]);
    got[k] is the string of read index first + k.  Returns the number of reads compared."""
    global _FULLSIZE
    if _FULLSIZE is None:
        _FULLSIZE = np.load(os.path.join(GOLDEN, "fullsize_digests.npz"))
    idx, ln, dig = _FULLSIZE[name + "_idx"], _FULLSIZE[name + "_len"], _FULLSIZE[name + "_dig"]
    n = 0
    for i, l, d in zip(idx.tolist(), ln.tolist(), dig.tolist()):
        k = i - first
        if 0 <= k < len(got):
            assert len(got[k]) == l and int(sha(got[k])[:16], 16) == d, (name, i)
            n += 1
    return n


@pytest.fixture(scope="module")
def ctx(tables):
    sub, nps = tables
    c = aln.Context(sub, nps, max_n=6, max_l=100, device=0)
    yield c
    c.close()


def test_library_is_gfx950_and_loaded():
    lib = _lib.load()
    assert lib.npore_device_count() >= 1


def test_dpp_directions():
    lib = _lib.load()
    out = (C.c_uint32 * 128)()
    assert lib.npore_debug_dpp(out) == 0
    prev, nxt = list(out[:64]), list(out[64:])
    assert prev[1:] == list(range(0, 63)) and prev[0] == 0
    assert nxt[:63] == list(range(1, 64)) and nxt[63] == 0


def test_unit_aligns(ctx):
    cases = load_json("unit_aligns.json")
    refs = [enc(c["ref"]) for c in cases]
    seqs = [enc(c["seq"]) for c in cases]
    cigs = [expand_cigar(c["cigar"]) for c in cases]
    a, st = ctx.align_batch(refs, seqs, cigs, max_b_rows=20, r=10, return_status=True)
    assert not st.any()
    assert a == [c["aln_20_10"] for c in cases]
    b = ctx.align_batch(refs, seqs, cigs)
    assert b == [c["aln_default"] for c in cases]


def test_reads_e2e_raw(ctx):
    fasta = "".join(l.strip() for l in open(os.path.join(GOLDEN, "data", "ref.fasta")) if not l.startswith(">")).upper()
    want = {r["name"]: r for r in load_json("reads_e2e.json")}
    names, refs, seqs, cigs = [], [], [], []
    for line in open(os.path.join(GOLDEN, "data", "reads.sam")):
        if line.startswith("@"):
            continue
        f = line.rstrip("\n").split("\t")
        ex = expand_cigar(f[5]).replace("S", "").replace("H", "")
        start = int(f[3]) - 1
        rlen = sum(1 for ch in ex if ch in "XD=M")
        names.append(f[0]); refs.append(enc(fasta[start:start + rlen])); seqs.append(enc(f[9])); cigs.append(ex)
    got, st = ctx.align_batch(refs, seqs, cigs, return_status=True)
    assert not st.any()
    for n, g in zip(names, got):
        assert g == want[n]["raw_align"], n


def test_synthetic_golden(ctx):
    recs = load_json("synthetic.json")
    groups = {}
    for r in recs:
        groups.setdefault((r["r"], r["max_b_rows"]), []).append(r)
    for (r, mbr), rs in groups.items():
        trip = [synth.make_pair(x["base_seed"], x["index"], x["ref_len"], x["p_np"], x["p_cnv"], x["mixed"]) for x in rs]
        got, st = ctx.align_batch([t[0] for t in trip], [t[1] for t in trip], [t[2] for t in trip],
                                  r=r, max_b_rows=mbr, return_status=True)
        assert not st.any(), (r, mbr)
        for x, g in zip(rs, got):
            assert len(g) == x["len"] and sha(g) == x["sha256"], x


@pytest.mark.parametrize("r", [1, 2, 5, 10, 30, 31, 32, 50, 63, 64, 100, 127, 128, 200])
def test_fuzz_vs_oracle(ctx, tables, r):
    sub, nps = tables
    rng = np.random.default_rng(1000 + r)
    refs, seqs, cigs = [], [], []
    for k in range(24):
        ref_len = int(rng.integers(1, 700))
        ref, seq, cig = synth.make_pair(500 + r, k, ref_len, float(rng.choice([0.0, 0.05, 0.15, 0.4])),
                                        float(rng.choice([0.0, 0.3, 0.9])))
        if k % 5 == 0 and len(ref) > 3:
            ref = ref.copy(); ref[rng.integers(0, len(ref), size=2)] = 0
        if k % 7 == 0 and len(seq) > 3:
            seq = seq.copy(); seq[rng.integers(0, len(seq), size=2)] = 0
        refs.append(ref); seqs.append(seq); cigs.append(cig)
    for mbr, ist, iex in ((20000, 5, 1), (64, 5, 1), (7, 5, 1), (2, 5, 1), (300, 3, 0), (300, 7, 2)):
        got, st = ctx.align_batch(refs, seqs, cigs, indel_start=ist, indel_extend=iex, r=r, max_b_rows=mbr,
                                  return_status=True)
        for k in range(len(refs)):
            want, wst = oracle.align(refs[k], seqs[k], cigs[k], sub, nps, indel_start=ist, indel_extend=iex,
                                     r=r, max_b_rows=mbr, return_status=True)
            assert got[k] == want and st[k] == wst, (r, mbr, ist, iex, k)


@pytest.mark.parametrize("r,chunks", [(100, 0), (100, 1), (100, 3), (100, 2), (30, 1), (30, 5), (30, 0),
                                      (40, 0), (70, 0), (70, 2), (95, 0), (96, 0), (140, 0), (170, 0), (200, 0),
                                      (230, 0), (230, 1)])
def test_kernel_shapes(tables, r, chunks):
    """Every waves-per-chunk count (1..8, following from r) and several chunks-per-workgroup
    packings give the same strings as the oracle."""
    sub, nps = tables
    c = aln.Context(sub, nps, max_n=6, max_l=100, device=0)
    c.set("force_chunks", chunks)
    refs, seqs, cigs = synth.make_batch(321, 12, ref_len=900, p_np=0.15)
    refs2, seqs2, cigs2 = synth.make_batch(322, 3, ref_len=2600, p_np=0.05)
    refs, seqs, cigs = refs + refs2, seqs + seqs2, cigs + cigs2
    for mbr in (20000, 333):
        got, st = c.align_batch(refs, seqs, cigs, r=r, max_b_rows=mbr, return_status=True)
        assert not st.any()
        for k in range(len(refs)):
            assert got[k] == oracle.align(refs[k], seqs[k], cigs[k], sub, nps, r=r, max_b_rows=mbr), (r, chunks, mbr, k)
    c.close()


def test_device_prep_matches_host_twin(ctx):
    """Path conversion, breaks, n-polymer annotation and word packing done by the prep
    kernels, word for word against tests/model/host_prep.hpp (itself checked vs the oracle)."""
    from model import model
    lib = _lib.load()
    rng = np.random.default_rng(4)
    desc_dt = np.dtype([("read_id", "i4"), ("brk", "i4"), ("nrows", "i4"), ("row0", "i4"), ("col0", "i4"),
                        ("drows", "i4"), ("dcols", "i4"), ("out_cap", "i4"), ("steps_off", "i8"), ("inss_off", "i8"),
                        ("seqw_off", "i8"), ("refw_off", "i8"), ("tb_off", "i8"), ("out_off", "i8"), ("seq_off", "i8"),
                        ("ref_off", "i8"), ("plain_lo", "i4"), ("plain_hi", "i4"), ("pad", "i4", 2)])
    assert desc_dt.itemsize == 112

    def fetch(what, dtype, count):
        a = np.zeros(count, dtype)
        assert lib.npore_debug_fetch(ctx.handle, what, a.ctypes.data, a.nbytes) == 0, _lib.last_error()
        return a

    for k in range(12):
        ref, seq, cig = synth.make_pair(900, k, int(rng.integers(50, 3000)), float(rng.choice([0.0, 0.1, 0.4])), 0.5)
        if k % 3 == 0:
            ref = ref.copy(); ref[rng.integers(0, len(ref), size=3)] = 0
        mbr = int(rng.choice([40, 333, 20000]))
        ctx.align_batch([ref], [seq], [cig], r=10, max_b_rows=mbr)
        want = model.prep(ref, seq, cig, max_b_rows=mbr)
        nch = int(fetch(7, np.int32, 2)[0])
        assert nch == want["n_chunks"]
        assert np.array_equal(fetch(0, np.uint8, len(want["steps"])), want["steps"])
        assert np.array_equal(fetch(1, np.int32, len(want["inss"])), want["inss"])
        d = fetch(2, desc_dt, nch)
        geom = np.stack([d[f] for f in ("brk", "nrows", "row0", "col0", "drows", "dcols", "out_cap")], axis=1)
        assert np.array_equal(geom, want["geom"])
        # plain range == the per-step predicate of cell.hpp (step_is_plain), evaluated row by row
        for q in range(nch):
            ins = want["inss"][d["brk"][q]:d["brk"][q] + d["nrows"][q]] - d["row0"][q]
            dl = np.arange(d["nrows"][q]) - ins
            plain = (ins - 10 >= 6) & (dl - 10 >= 6) & (ins + 10 <= d["drows"][q]) & (dl + 10 <= d["dcols"][q])
            idx = np.nonzero(plain)[0]
            if len(idx):
                assert (d["plain_lo"][q], d["plain_hi"][q]) == (idx[0], idx[-1] + 1) and plain[idx[0]:idx[-1] + 1].all()
            else:
                assert d["plain_lo"][q] >= d["plain_hi"][q]
        assert np.array_equal(fetch(3, np.uint32, len(want["seqw"])), want["seqw"])
        assert np.array_equal(fetch(4, np.uint32, want["refw"].size).reshape(-1, 4), want["refw"])
        got_l = fetch(5, np.uint8, want["refl"].size).reshape(-1, 8)
        assert np.array_equal(got_l[:, :6], want["refl"][:, :6])
        sched = fetch(6, np.int32, nch)
        assert sorted(sched.tolist()) == list(range(nch))
        assert all(d["nrows"][sched[i]] >= d["nrows"][sched[i + 1]] for i in range(nch - 1))


def test_get_np_info_device(ctx):
    seqs = [enc(s) for s in load_json("np_info_seqs.json") if len(s)]
    rng = np.random.default_rng(8)
    for k in range(20):
        seqs.append(rng.choice(rng.integers(0, 5, size=int(rng.integers(1, 4))), size=int(rng.integers(1, 3000))).astype(np.uint8))
    for s in seqs:
        assert np.array_equal(ctx.get_np_info(s), oracle.get_np_info(s))


def test_get_np_info_segments_of_long_sequences(tables):
    """get_np_info() of sequences far longer than one wave's segment (annot_wave.hpp np_info_wave_kernel: 16 384 positions
    per wave, each wave warmed up on the positions in front of its segment): arrays of every period that cross the segment
    boundaries, nested periods, two-letter sequences (periodic everywhere), N stretches -- every value equal to the
    oracle's, at the default table shape and at other max_n / max_l."""
    rng = np.random.default_rng(12)
    n = 70_000

    def make(kind):
        if kind == 0:
            parts = []
            while sum(map(len, parts)) < n:
                per = int(rng.integers(1, 7))
                parts.append(np.tile(rng.integers(1, 5, per).astype(np.uint8), int(rng.integers(3, 400))))
                parts.append(rng.integers(1, 5, int(rng.integers(0, 6))).astype(np.uint8))
            s = np.concatenate(parts)[:n]
        else:
            s = rng.integers(1, 3, n).astype(np.uint8)
        for b in (16384, 32768, 49152):              # arrays planted across every segment boundary, at every phase
            per = int(rng.integers(1, 7))
            a = b - int(rng.integers(1, 900))
            s[a:a + 1000] = np.tile(s[a:a + per], 1000 // per + 1)[:1000]
        for _ in range(4):
            a = int(rng.integers(0, n - 50))
            s[a:a + int(rng.integers(1, 40))] = 0
        return s

    for max_n, max_l in ((6, 100), (6, 127), (4, 20), (3, 5)):
        c = aln.Context(None, None, max_n=max_n, max_l=max_l, device=0)            # an annotation-only context
        for kind in (0, 1):
            s = make(kind)
            got, want = c.get_np_info(s), np.asarray(oracle.get_np_info(s, max_n=max_n, max_l=max_l))
            assert np.array_equal(got, want), (max_n, max_l, kind, np.argwhere(got != want)[:4])
        c.close()


def test_large_max_b_rows(ctx, tables):
    """max_b_rows above the default: one chunk longer than the LDS-resident annotation
    planes cover (global-scratch planes), 16-bit run lengths near their range."""
    sub, nps = tables
    refs, seqs, cigs = synth.make_batch(77, 2, ref_len=16_000)
    for mbr in (40000, 60000, 25000):
        got, st = ctx.align_batch(refs, seqs, cigs, r=30, max_b_rows=mbr, return_status=True)
        assert not st.any()
        for k in range(2):
            assert got[k] == oracle.align(refs[k], seqs[k], cigs[k], sub, nps, r=30, max_b_rows=mbr), (mbr, k)
    # chunk slices longer than gather_kernel stages in LDS (> 24 K bases): its global-memory path, several output tiles
    refs2, seqs2, cigs2 = synth.make_batch(78, 1, ref_len=30_000)
    for mbr in (40000, 60000):
        got, st = ctx.align_batch(refs2, seqs2, cigs2, r=30, max_b_rows=mbr, return_status=True)
        assert not st.any()
        assert got[0] == oracle.align(refs2[0], seqs2[0], cigs2[0], sub, nps, r=30, max_b_rows=mbr), mbr
    with pytest.raises(aln.NporeError):
        ctx.align_batch(refs, seqs, cigs, r=30, max_b_rows=70000)     # refused loudly, not silently wrong
    with pytest.raises(aln.NporeError):
        ctx.align_batch(refs, seqs, cigs, r=512)


def test_groups_under_small_traceback_budget(tables):
    """A batch whose traceback does not fit the budget is processed in several groups of reads
    (work buffers reused); results and order are unchanged."""
    sub, nps = tables
    c = aln.Context(sub, nps, max_n=6, max_l=100, device=0)
    refs, seqs, cigs = synth.make_batch(55, 40, ref_len=2500)
    want = c.align_batch(refs, seqs, cigs, r=30)
    c.set("tb_budget_mb", 8)            # ~ 5 reads of 2.5 kb at r=30 per group
    got, st = c.align_batch(refs, seqs, cigs, r=30, return_status=True)
    assert not st.any() and got == want
    assert c.timing()["launches"] >= 4
    for k in (0, 39):
        assert got[k] == oracle.align(refs[k], seqs[k], cigs[k], sub, nps, r=30)
    # the same through the device glue: every group standardises its own reads
    from npore_amd import cig
    fin, st = c.align_batch(refs, seqs, cigs, r=30, return_status=True, final_cigars=True)
    assert not st.any() and fin == cig.standardize_batch(want, refs, seqs)
    assert c.timing()["launches"] >= 4
    c.close()


def test_widest_band(ctx, tables):
    sub, nps = tables
    refs, seqs, cigs = synth.make_batch(66, 3, ref_len=1200, p_np=0.1)
    got, st = ctx.align_batch(refs, seqs, cigs, r=255, return_status=True)
    assert not st.any()
    for k in range(3):
        assert got[k] == oracle.align(refs[k], seqs[k], cigs[k], sub, nps, r=255)


@pytest.mark.parametrize("r", [256, 300, 384, 447, 511])
def test_bands_of_nine_to_sixteen_waves(ctx, tables, r):
    """r = 256 ... 511: 9 ... 16 waves per chunk, one chunk per workgroup, through the instantiation of the fill kernel that
    takes its wave count from the launch (and the four-load rows of the row-per-hop traceback): strings and status
    bits equal the oracle's, also with many short chunks and with reads whose input path is far from the best one."""
    sub, nps = tables
    refs, seqs, cigs = synth.make_batch(600 + r, 5, ref_len=1100, p_np=0.1)
    ref, seq = refs[0], seqs[0]
    m = min(len(ref), len(seq)) - 30
    refs.append(ref); seqs.append(seq); cigs.append("I" * (len(seq) - m) + "D" * (len(ref) - m) + "=" * m)
    for mbr in (1500, 150):
        got, st = ctx.align_batch(refs, seqs, cigs, r=r, max_b_rows=mbr, return_status=True)
        for k in range(len(refs)):
            want, wst = oracle.align(refs[k], seqs[k], cigs[k], sub, nps, r=r, max_b_rows=mbr, return_status=True)
            assert got[k] == want and st[k] == wst, (r, mbr, k)


def test_div_small_domain():
    """The device's float-reciprocal division is exact on its whole domain."""
    lib = _lib.load()
    bad = C.c_int64(-1)
    assert lib.npore_debug_divcheck(C.byref(bad)) == 0
    assert bad.value == 0


def test_edge_inputs(ctx):
    got, st = ctx.align_batch([enc(""), enc("ACGT"), enc(""), enc("ACGT"), enc("ACGT")],
                              [enc(""), enc(""), enc("ACGT"), enc("ACGT"), enc("ACGT")],
                              ["", "DDDD", "IIII", "===", "==N="], return_status=True)
    assert got[:3] == ["", "DDDD", "IIII"]
    assert st[:3].tolist() == [0, 0, 0]
    assert st[3] & 32 and st[4] & 32
    assert ctx.align_batch([], [], []) == []


def test_align_signature_drop_in(tables):
    sub, nps = tables
    c = load_json("unit_aligns.json")[8]
    s = aln.align(enc(c["ref"]), enc(c["seq"]), expand_cigar(c["cigar"]), sub, nps, max_b_rows=20, r=10)
    assert s == c["aln_20_10"]
    info = aln.get_np_info(enc("ATATATATTTTTTAAAGCGCGC"))
    assert info[:, 0, 0].tolist() == [0, 0, 0, 0, 0, 0, 0, 6, 6, 6, 6, 6, 6, 3, 3, 3, 0, 0, 0, 0, 0, 0]


def test_10kb_properties(ctx, tables):
    """Full-size reads: size-independent properties (ops consume exactly the read
    and the reference; '=' / 'X' agree with the bases) plus oracle equality on 2."""
    sub, nps = tables
    refs, seqs, cigs = synth.make_batch(2, 32, ref_len=10_000)
    got, st = ctx.align_batch(refs, seqs, cigs, r=100, return_status=True)
    assert not st.any()
    for ref, seq, g in zip(refs, seqs, got):
        ops = np.frombuffer(g.encode(), np.uint8)
        assert np.isin(ops, [ord("="), ord("X"), ord("I")]).sum() == len(seq)
        assert np.isin(ops, [ord("="), ord("X"), ord("D")]).sum() == len(ref)
        i = j = 0
        for op in g:
            if op == "=":
                assert ref[j] == seq[i]; i += 1; j += 1
            elif op == "X":
                assert ref[j] != seq[i]; i += 1; j += 1
            elif op == "I":
                i += 1
            else:
                j += 1
    for k in (0, 17):
        assert got[k] == oracle.align(refs[k], seqs[k], cigs[k], sub, nps, r=100)
    assert check_fullsize_digests("c2", got) == 32


def test_c2_every_read_bit_exact(ctx):
    """BASELINE.json configs[1] (1 000 reads of 10 kb, seed 2, r=100): EVERY read against the pinned
    oracle's digest."""
    refs, seqs, cigs = synth.make_batch(2, 1000)
    got, st = ctx.align_batch(refs, seqs, cigs, r=100, return_status=True)
    assert not st.any()
    assert check_fullsize_digests("c2", got) == 1000


def _check_alignment_properties(ref, seq, g):
    """Size-independent properties of one output string, vectorised: the ops consume exactly the read and the
    reference, '=' pairs equal bases and 'X' pairs different ones (src/aln.pyx:732-735)."""
    ops = np.frombuffer(g.encode(), np.uint8)
    uses_seq = (ops == ord("=")) | (ops == ord("X")) | (ops == ord("I"))
    uses_ref = (ops == ord("=")) | (ops == ord("X")) | (ops == ord("D"))
    assert (uses_seq | uses_ref).all()                       # nothing but = X I D
    assert int(uses_seq.sum()) == len(seq) and int(uses_ref.sum()) == len(ref)
    i = np.cumsum(uses_seq) - uses_seq                       # read / reference position each op starts at
    j = np.cumsum(uses_ref) - uses_ref
    eq, x = ops == ord("="), ops == ord("X")
    assert (ref[j[eq]] == seq[i[eq]]).all() and (ref[j[x]] != seq[i[x]]).all()


def test_c3_all_distinct_10000_reads_one_call(ctx, tables):
    """BASELINE.json configs[2] at a tenth of its read count but full read size, ALL DISTINCT, in
    ONE library call: 10 000 reads of 10 kb from the seed-3 mixed-density generator at r=100 (163 GB of
    traceback words: the library splits the call into groups by its memory budget).  Size-independent properties
    on every read, oracle equality on two of each n-polymer density."""
    import multiprocessing as mp
    sub, nps = tables
    n, span = 10_000, 500
    with mp.get_context("spawn").Pool(8) as pool:            # fresh workers: this process holds a live HIP runtime
        parts = pool.map(synth.make_span, [(3, span, 10_000, True, k, 1) for k in range(0, n, span)])
    refs = [x for p in parts for x in p[0]]; seqs = [x for p in parts for x in p[1]]; cigs = [x for p in parts for x in p[2]]
    assert len(refs) == n
    got, st = ctx.align_batch(refs, seqs, cigs, r=100, return_status=True)
    assert not st.any() and len(got) == n
    assert ctx.timing()["launches"] >= 1
    for ref, seq, g in zip(refs, seqs, got):
        _check_alignment_properties(ref, seq, g)
    # the mixed generator draws p_np per read from {0, 0.02, 0.05, 0.15}: reads with few and with many n-polymers
    dens = np.array([(np.diff(r_) == 0).mean() for r_ in refs[:400]])
    pick = list(np.argsort(dens)[[0, 1, -2, -1]])
    for k in pick:
        assert got[k] == oracle.align(refs[k], seqs[k], cigs[k], sub, nps, r=100), k
    assert check_fullsize_digests("c3", got) == 10_000           # every read against the pinned oracle's digest
    # a batch is a function of its reads only: the same reads in a small call give the same strings
    again = ctx.align_batch(refs[4990:5010], seqs[4990:5010], cigs[4990:5010], r=100)
    assert again == got[4990:5010]


def test_c3_full_100000_reads_one_call(ctx, tables):
    """BASELINE.json configs[2] at its FULL size: 100 000 all-distinct reads of 10 kb from the seed-3
    mixed-density generator at r=100 in ONE library call (3.1 GB of bases + CIGARs in, 1.1 GB of strings out, 1.6 TB
    of traceback words over the call -- a score of groups through the two work sets, each with traceback offsets far
    beyond 2^32 words).  Size-independent properties on EVERY read, oracle equality on the reads of the lowest and
    highest n-polymer density among the first 400, and a sub-batch from beyond the 4 GB mark re-run on its own."""
    import multiprocessing as mp
    from concurrent.futures import ThreadPoolExecutor
    sub, nps = tables
    n, span = 100_000, 1000
    with mp.get_context("spawn").Pool(min(16, os.cpu_count() or 1)) as pool:   # fresh workers: this process holds a live HIP runtime
        parts = pool.map(synth.make_span, [(3, span, 10_000, True, k, 1) for k in range(0, n, span)], chunksize=1)
    refs = [x for p in parts for x in p[0]]; seqs = [x for p in parts for x in p[1]]; cigs = [x for p in parts for x in p[2]]
    del parts
    assert len(refs) == n
    got, st = ctx.align_batch(refs, seqs, cigs, r=100, return_status=True)
    assert not st.any() and len(got) == n
    assert sum(len(c) for c in cigs) > 10**9 and sum(len(g) for g in got) > 10**9

    def check(lo):
        for k in range(lo, min(lo + 500, n)):
            _check_alignment_properties(refs[k], seqs[k], got[k])
        return True
    with ThreadPoolExecutor(8) as tp:          # numpy releases the GIL in the reductions
        assert all(tp.map(check, range(0, n, 500)))
    dens = np.array([(np.diff(r_) == 0).mean() for r_ in refs[:400]])
    for k in list(np.argsort(dens)[[0, -1]]) + [n - 1]:
        assert got[k] == oracle.align(refs[k], seqs[k], cigs[k], sub, nps, r=100), k
    # the first 10 000 reads and every 10th after them, bit for bit against the pinned oracle's digests
    assert check_fullsize_digests("c3", got) == 19_000
    again = ctx.align_batch(refs[99_000:99_020], seqs[99_000:99_020], cigs[99_000:99_020], r=100)
    assert again == got[99_000:99_020]


def test_production_default_r30_one_full_launch(ctx, tables):
    """The tool's default band (reference src/realign.py:46-51: r=30, max_b_rows=20000) at exactly one full launch of
    the fill kernel (4 000 reads of 10 kb = 4 096 resident single-wave chunks): properties on every read, oracle on four."""
    sub, nps = tables
    refs, seqs, cigs = synth.make_batch(2, 4000)
    got, st = ctx.align_batch(refs, seqs, cigs, r=30, return_status=True)
    assert not st.any()
    for ref, seq, g in zip(refs, seqs, got):
        _check_alignment_properties(ref, seq, g)
    for k in (0, 1333, 2666, 3999):
        assert got[k] == oracle.align(refs[k], seqs[k], cigs[k], sub, nps, r=30), k
    assert check_fullsize_digests("r30", got) == 4000            # EVERY read against the pinned oracle's digest


@pytest.mark.parametrize("r,n,reps", [(40, 1000, 16), (100, 1000, 16), (120, 1000, 16), (140, 1000, 16), (200, 1000, 16), (30, 4000, 8)])
def test_full_launch_repeated_is_identical(ctx, tables, r, n, reps):
    """A full launch of 10 kb reads (every chunk slot busy at once; 1, 2, 4, 5 and 7 waves per chunk), several launches in
    a row: every launch must give the strings of the first, and the first the oracle's on four reads.  Guards what
    depends on timing -- the waves' hand-shake and the hazards between the step assembly and the compiled code around
    it: a slip there shows as a read or two per thousand with a wrong stretch, in some launches only (two such hazards
    were found and fixed in round 3 with an experimental wave placement)."""
    sub, nps = tables
    refs, seqs, cigs = synth.make_batch(2, n)
    first, st = ctx.align_batch(refs, seqs, cigs, r=r, return_status=True)
    assert not st.any()
    for k in (0, n // 3, 2 * n // 3, n - 1):
        assert first[k] == oracle.align(refs[k], seqs[k], cigs[k], sub, nps, r=r), k
    for rep in range(reps - 1):
        got = ctx.align_batch(refs, seqs, cigs, r=r)
        bad = [k for k in range(len(got)) if got[k] != first[k]]
        assert not bad, (rep, bad[:8])


def test_chunkmajor_placement_matches_shipped(ctx, tables, tmp_path):
    """The wave placement that EXPOSED the two inline-assembly hazards of round 3: with
    -DNPORE_X_CHUNKMAJOR the four waves of a chunk sit on four SIMDs and each issues back to back while its neighbours
    wait -- the only configuration in which a hazard of the step assembly (a late load into a scratch register, a VALU
    write racing a ds_write_b128's data read) has ever shown, as one or two reads per thousand with a wrong stretch in
    every third launch.  24 full launches of C2's 1 000 reads at r = 100 and r = 120 with that build (made by
    __graft_entry__.build(), or here when it is missing) must give the shipped placement's strings, every launch."""
    import json
    import subprocess
    import sys
    from conftest import REPO
    lib = os.path.join(REPO, "tests", "model", "libnpore_amd_chunkmajor.so")
    _lib.build(defines=("NPORE_X_CHUNKMAJOR",), out=lib)
    out = tmp_path / "cm.json"
    subprocess.check_call([sys.executable, os.path.join(REPO, "tests", "tools", "placement_check.py"), str(out), "24", "100", "120"],
                          env=dict(os.environ, NPORE_AMD_LIB=lib), cwd=REPO)
    res = json.load(open(out))
    refs, seqs, cigs = synth.make_batch(2, 1000)
    for r in (100, 120):
        want = ctx.align_batch(refs, seqs, cigs, r=r)
        assert [sha(s)[:16] for s in want] == res[str(r)]["first"], r
        bad = [(rep, d[:4]) for rep, d in enumerate(res[str(r)]["differ"]) if d]
        assert not bad, (r, bad[:4])


def test_fuzz_time_boxed():
    """A time-boxed leg of tests/tools/fuzz_gpu.py under the driver (fixed seeds; ~20 s over all shapes, ~25 s focused on
    what has failed before: 4-8 waves per chunk with chunk heights of 2...64 anti-diagonals, tables with max_l < 32):
    every string and status equal to the oracle's."""
    import importlib.util
    from conftest import REPO
    spec = importlib.util.spec_from_file_location("fuzz_gpu", os.path.join(REPO, "tests", "tools", "fuzz_gpu.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    msgs = []
    total = 0
    for seed, focus, secs in ((301, False, 20.0), (302, True, 25.0)):
        rounds, reads, bad = fz.fuzz(secs, seed, focus=focus, log=msgs.append)
        assert bad == 0, msgs[:5]
        assert rounds >= 3 and reads >= 30, (seed, rounds, reads)
        total += reads
    print(f"fuzz leg: {total} reads")


def test_fuzz_pack_time_boxed(ctx):
    """A time-boxed leg of tests/tools/fuzz_pack.py: random small BAMs (1 ... 60 records of 1 ... 3 000 bases, clips, ambiguity
    codes, batches of 1 ... 64 reads) through the file pipeline three ways -- inputs unpacked and texts compacted on the device,
    packed on the host, packed and standardised on the host -- byte-identical SAM and status.  (The first run of the tool on
    the text compaction failed in its first round: batches whose slots were smaller than the part sent ahead.)"""
    import importlib.util
    from conftest import REPO
    spec = importlib.util.spec_from_file_location("fuzz_pack", os.path.join(REPO, "tests", "tools", "fuzz_pack.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    msgs = []
    rounds, reads, clean, bad = fz.fuzz(15.0, 77, ctx, log=msgs.append)
    assert bad == 0, msgs[:5]
    assert rounds >= 5 and reads >= 50 and clean >= 20, (rounds, reads, clean)


def test_c5_256_ultralong_reads_r200(ctx, tables):
    """BASELINE.json configs[4] at full size: 256 reads of 50 kb, r=200 (7 waves per chunk,
    6 chunks per read, 41 GB of traceback words), properties on every read + oracle equality on two."""
    sub, nps = tables
    refs, seqs, cigs = synth.make_batch(5, 256, ref_len=50_000)
    got, st = ctx.align_batch(refs, seqs, cigs, r=200, return_status=True)
    assert not st.any()
    for ref, seq, g in zip(refs, seqs, got):
        _check_alignment_properties(ref, seq, g)
    for k in (3, 200):
        assert got[k] == oracle.align(refs[k], seqs[k], cigs[k], sub, nps, r=200), k
    assert check_fullsize_digests("c5", got) == 256              # EVERY read against the pinned oracle's digest


def test_realign_cli_end_to_end(tmp_path):
    """BAM in -> SAM out through `python -m npore_amd.realign` == the reference's own golden
    output test/data/npore_realigned.sam (record by record; the @PG line differs by design)."""
    import subprocess
    import sys
    from conftest import REPO
    prefix = str(tmp_path / "realigned")
    subprocess.check_call([sys.executable, "-m", "npore_amd.realign", "--bam", os.path.join(GOLDEN, "data", "reads.bam"),
                           "--ref", os.path.join(GOLDEN, "data", "ref.fasta"), "--out_prefix", prefix], cwd=REPO)
    # the pure-Python reader / writer (the restatement the native host I/O is tested against) gives the same file
    subprocess.check_call([sys.executable, "-m", "npore_amd.realign", "--bam", os.path.join(GOLDEN, "data", "reads.bam"),
                           "--ref", os.path.join(GOLDEN, "data", "ref.fasta"), "--out_prefix", prefix + "_py", "--python_io"],
                          cwd=REPO)
    body = lambda path: [l for l in open(path) if not l.startswith("@PG")]
    assert body(prefix + ".sam") == body(prefix + "_py.sam")
    # (the default run above read the BAM in ONE PASS; the indexed reader, which several ranks and BED regions use, writes the same)
    subprocess.check_call([sys.executable, "-m", "npore_amd.realign", "--bam", os.path.join(GOLDEN, "data", "reads.bam"),
                           "--ref", os.path.join(GOLDEN, "data", "ref.fasta"), "--out_prefix", prefix + "_ix"],
                          cwd=REPO, env=dict(os.environ, NPORE_BAM_ONE_PASS="0"))
    assert body(prefix + ".sam") == body(prefix + "_ix.sam")

    def records(path):
        hdr, recs = [], {}
        for line in open(path):
            if line.startswith("@"):
                hdr.append(line.rstrip("\n"))
            else:
                f = line.rstrip("\n").split("\t")
                recs[f[0]] = f
        return hdr, recs

    hdr, got = records(prefix + ".sam")
    ghdr, want = records(os.path.join(GOLDEN, "data", "npore_realigned.sam"))
    assert hdr[:2] == ghdr[:2] and hdr[2].startswith("@PG\tPN:realigner\tID:realigner\tVN:")
    assert got.keys() == want.keys() and len(got) == 10
    for name in want:
        assert got[name] == want[name], name


def test_native_realign_batch_matches_python_pipeline(ctx, tmp_path):
    """BAM records -> SAM text through npore_bam_realign_batch == the Python pipeline (get_read_data ->
    align_batch -> standardize_batch -> sam_line) on a synthetic BAM with soft and hard clips, N / ambiguity codes, both
    strands, HP tags and a read whose CIGAR disagrees with its sequence (refused, not written)."""
    import argparse
    from npore_amd import bam, cfg
    from npore_amd.cig import bases_to_int, expand_cigar, standardize_batch
    rng = np.random.default_rng(5)
    refs, seqs, cigs = synth.make_batch(77, 24, ref_len=1500, p_np=0.1)
    dec = lambda a: "".join("NACGT"[x] for x in a)
    contig, pos0, recs = [], [], []
    for k, (rf, sq, cg) in enumerate(zip(refs, seqs, cigs)):
        pos0.append(len(contig) + 20)
        contig += list("ACGT"[x] for x in rng.integers(0, 4, 20)) + list(dec(rf))
        cg = cg.decode() if isinstance(cg, (bytes, bytearray)) else "".join(chr(x) for x in cg) if not isinstance(cg, str) else cg
        runs, last, cnt = [], None, 0
        for ch in cg:
            if ch == last: cnt += 1
            else:
                if last is not None: runs.append(("MIDNSHP=XB".index(last), cnt))
                last, cnt = ch, 1
        runs.append(("MIDNSHP=XB".index(last), cnt))
        lead, trail = (3 if k % 2 else 0), (2 if k % 3 == 0 else 0)
        cig = ([(4, lead)] if lead else []) + runs + ([(4, trail)] if trail else [])
        if k % 6 == 2:
            cig = [(5, 4)] + cig                    # hard clips outside the soft ones (or alone)
        if k % 4 == 3:
            cig = cig + [(5, 7)]
        body = dec(sq)
        if k % 7 == 3:                              # N in the read, an ambiguity code (both are base code 0)
            body = body[:11] + "N" + body[12:40] + "R" + body[41:]
        recs.append(dict(name=f"r{k}", flag=16 if k % 4 == 1 else 0, ref_id=0, pos=pos0[-1], cigar=cig,
                         seq="A" * lead + body + "C" * trail, qual=None if k % 5 == 0 else bytes([30]) * (lead + len(sq) + trail),
                         hp=k % 3))
    recs[7]["cigar"] = recs[7]["cigar"][:-1] + [(0, 5)] if recs[7]["cigar"][-1][0] != 4 else recs[7]["cigar"] + [(0, 5)]   # lengths now disagree
    contig = "".join(contig) + "ACGT" * 10
    (tmp_path / "c.fa").write_text(">ctg\n" + contig + "\n")
    bam.write_bam(str(tmp_path / "s.bam"), [("ctg", len(contig))], recs)
    old = cfg.args
    cfg.args = argparse.Namespace(max_n=6, max_l=100, regions=[("ctg", 0, len(contig) - 1)], max_reads=0)
    try:
        nb, nf = bam.NativeBam(str(tmp_path / "s.bam")), bam.NativeFasta(str(tmp_path / "c.fa"))
        idx = nb.select(cfg.args.regions)
        text, st = nb.realign_batch(ctx, nf, idx, r=30)
        text = bytes(text)
        py = bam.BamFile(str(tmp_path / "s.bam"))
        rds = list(bam.get_read_data(py, bam.read_fasta(str(tmp_path / "c.fa"))))
        pc = [expand_cigar(rd[5]).replace("S", "").replace("H", "") for rd in rds]
        pr, ps = [bases_to_int(rd[9]) for rd in rds], [bases_to_int(rd[7]) for rd in rds]
        alns, pst = ctx.align_batch(pr, ps, pc, r=30, return_status=True)
        finals = standardize_batch(alns, pr, ps)
        want = "".join(bam.sam_line(rd, f) for rd, f, s_ in zip(rds, finals, pst) if not s_ & 32)
        assert np.array_equal(st, pst) and ((st & 32) != 0).sum() == 1
        assert text.decode() == want and text.count(b"\n") == len(rds) - 1
        # the library's own overlapped batch loop (pack k+1 | GPU k | format + write k-1) writes the same file
        out = tmp_path / "pipe.sam"
        st2 = nb.realign_file(ctx, nf, idx, str(out), batch_reads=5, r=30)
        assert np.array_equal(st2, pst) and out.read_bytes() == text
        # (that loop hands the device the HEADS of the records and unpacks align()'s inputs there, unpack_kernels.hpp;
        # with the pack on the host, and with the glue on the host as well, the file is the same)
        for key in ("device_pack", "device_glue"):
            ctx.set(key, 0)
            outh = tmp_path / f"pipe_no_{key}.sam"
            sth = nb.realign_file(ctx, nf, idx, str(outh), batch_reads=5, r=30)
            assert np.array_equal(sth, pst) and outh.read_bytes() == text, key
        ctx.set("device_pack", 1)
        ctx.set("device_glue", 1)
        # another FASTA of the same size through the same context: the device copy follows the FASTA, not its size or address
        (tmp_path / "c2.fa").write_text(">ctg\n" + contig[::-1] + "\n")
        nf2 = bam.NativeFasta(str(tmp_path / "c2.fa"))
        out_d, out_h = tmp_path / "other_fa_dev.sam", tmp_path / "other_fa_host.sam"
        st_d = nb.realign_file(ctx, nf2, idx, str(out_d), batch_reads=5, r=30)
        ctx.set("device_pack", 0)
        st_h = nb.realign_file(ctx, nf2, idx, str(out_h), batch_reads=5, r=30)
        ctx.set("device_pack", 1)
        assert np.array_equal(st_d, st_h) and out_d.read_bytes() == out_h.read_bytes() and out_d.read_bytes() != text
        st_b = nb.realign_file(ctx, nf, idx, str(tmp_path / "back.sam"), batch_reads=5, r=30)
        assert np.array_equal(st_b, pst) and (tmp_path / "back.sam").read_bytes() == text
        # a traceback budget that cuts every batch into several groups of reads: each group uploads and unpacks its own slice
        ctx.set("tb_budget_mb", 2)
        try:
            st_g = nb.realign_file(ctx, nf, idx, str(tmp_path / "groups.sam"), batch_reads=12, r=30)
            assert np.array_equal(st_g, pst) and (tmp_path / "groups.sam").read_bytes() == text
        finally:
            ctx.set("tb_budget_mb", 0)
        # a STREAMED handle (bounded-memory ingest: every batch inflates the BGZF blocks its records lie in) writes the same
        ns = bam.NativeBam(str(tmp_path / "s.bam"), stream=True)
        assert ns.streamed and np.array_equal(ns.select(cfg.args.regions), idx)
        out3 = tmp_path / "pipe_streamed.sam"
        st3 = ns.realign_file(ctx, nf, idx, str(out3), batch_reads=5, r=30)
        assert np.array_equal(st3, pst) and out3.read_bytes() == text
        text4, st4 = ns.realign_batch(ctx, nf, idx[::-1], r=30)
        assert np.array_equal(st4, pst[::-1]) and sorted(bytes(text4).splitlines()) == sorted(text.splitlines())
        ns.close()
        # ONE PASS over the file (header-only handle, no record index; every block inflated once, records filtered as they
        # go by): the same bytes, the refused read reported by its ordinal -- also with one-block windows, so that nearly
        # every record straddles two windows
        for win in (None, "1"):
            if win:
                os.environ["NPORE_BAM_WINDOW_BLOCKS"] = win
            try:
                no = bam.NativeBam(str(tmp_path / "s.bam"), one_pass=True)
                assert no.one_pass and no.references == nb.references and no.lengths == nb.lengths and len(no.select(cfg.args.regions)) == 0
                out5 = tmp_path / f"onepass{win}.sam"
                n5, bad5, (refused5, incons5) = no.realign_sequential(ctx, nf, cfg.args.regions, str(out5), batch_reads=5, r=30)
                assert out5.read_bytes() == text and n5 == len(rds) and (refused5, incons5) == (1, 0) and bad5 == [(7, 32)]
                out6 = tmp_path / f"onepass_cap{win}.sam"
                n6, _, _ = no.realign_sequential(ctx, nf, cfg.args.regions, str(out6), batch_reads=4, max_reads=9, r=30)
                assert n6 == 9 and out6.read_bytes() == b"".join(text.splitlines(keepends=True)[:8])      # (read 7 of the 9 is refused)
                with pytest.raises(bam.OnePassUnsupported):
                    no.realign_sequential(ctx, nf, cfg.args.regions * 2, str(tmp_path / "x.sam"), r=30)
                no.close()
            finally:
                os.environ.pop("NPORE_BAM_WINDOW_BLOCKS", None)
    finally:
        cfg.args = old


def test_realign_hap_long_sequences(ctx, tables):
    """realign_hap's use of align(): whole haplotype-length sequences (tens of chunks each) --
    raw strings equal the oracle's, final CIGARs equal the Python standardisation of them."""
    from npore_amd import bam
    from npore_amd.cig import standardize
    sub, nps = tables
    dec = lambda a: "".join("NACGT"[x] for x in a)
    haps = []
    for k, L in enumerate((150_000, 61_000)):
        ref, seq, cig = synth.make_pair(91, k, L, 0.05, 0.3, False)
        haps.append(("chrT", k + 1, dec(seq), dec(ref), cig.decode() if isinstance(cig, bytes) else cig))
    out = bam.realign_haps(ctx, haps, r=30)
    for h, o in zip(haps, out):
        r_, s_ = np.frombuffer(h[3].encode().translate(bytes.maketrans(b"NACGT", bytes(range(5)))), np.uint8), \
                 np.frombuffer(h[2].encode().translate(bytes.maketrans(b"NACGT", bytes(range(5)))), np.uint8)
        want_raw = oracle.align(r_, s_, h[4], sub, nps, r=30)
        assert ctx.align_batch([r_], [s_], [h[4]], r=30)[0] == want_raw
        assert o[:4] == h[:4] and o[4] == standardize(want_raw, r_, s_)


def test_one_pass_shares_tile_the_file(ctx, tmp_path):
    """Several ranks, each in ONE pass over its stretch of the file (npore_bam_set_share: the record stream cut at virtual
    offsets of the .bai linear index): the stretches of 1 / 2 / 3 / 5 ranks begin at record starts, and the ranks' outputs
    concatenated in rank order are byte for byte the single process's file -- also with one-block windows (nearly every
    record straddles two) and with a rank that gets nothing; no .bai: refused, and `realign` under torch.distributed.run
    (2 ranks on this box's card) writes the same records in the same order as one process."""
    import argparse
    import subprocess
    import sys
    from conftest import REPO
    from npore_amd import bam, cfg
    sys.path.insert(0, os.path.join(REPO, "scripts"))
    import bench_realign
    bp, fa, clen = bench_realign.build_inputs(str(tmp_path), 260, 0, 3000, 17, procs=2)
    old = cfg.args
    cfg.args = argparse.Namespace(max_n=6, max_l=100, regions=[("ctg", 0, clen - 1)], max_reads=0)
    try:
        nf = bam.NativeFasta(fa)
        one = bam.NativeBam(bp, one_pass=True)
        with pytest.raises(bam.OnePassUnsupported):
            one.set_share(0, 2)                                       # no .bai yet
        whole = tmp_path / "whole.sam"
        n_all, bad, _ = one.realign_sequential(ctx, nf, cfg.args.regions, str(whole), batch_reads=50, r=30)
        assert n_all == 260 and not bad
        one.close()
        bam.write_bai(bp)
        for world, win in ((1, None), (2, None), (3, "1"), (5, None), (64, "2")):
            if win:
                os.environ["NPORE_BAM_WINDOW_BLOCKS"] = win
            try:
                parts, total, empty = [], 0, 0
                for rank in range(world):
                    h = bam.NativeBam(bp, one_pass=True, share=False)
                    has, b0, e0, _ = h.set_share(rank, world)
                    out = tmp_path / f"w{world}_r{rank}.sam"
                    n, bad, _ = h.realign_sequential(ctx, nf, cfg.args.regions, str(out), batch_reads=37, r=30)
                    h.close()
                    assert not bad and (has == 1) == (world > 1)
                    parts.append(out.read_bytes())
                    total += n
                    empty += n == 0
                assert total == 260 and b"".join(parts) == whole.read_bytes(), world
                assert empty == 0 if world <= 5 else empty > 0       # (64 ranks on 55 index windows: some get nothing)
            finally:
                os.environ.pop("NPORE_BAM_WINDOW_BLOCKS", None)
        # max_reads needs one process
        h = bam.NativeBam(bp, one_pass=True, share=False)
        h.set_share(1, 2)
        with pytest.raises(bam.OnePassUnsupported):
            h.realign_sequential(ctx, nf, cfg.args.regions, str(tmp_path / "x.sam"), max_reads=5, r=30)
        h.close()
        nf.close()
    finally:
        cfg.args = old
    # the tool itself: one process, then two ranks (each one pass over its stretch: the part files are in file order)
    recs = lambda path: [l for l in open(path) if not l.startswith("@")]
    subprocess.check_call([sys.executable, "-m", "npore_amd.realign", "--bam", bp, "--ref", fa, "--out_prefix", str(tmp_path / "cli1")], cwd=REPO)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(29300 + os.getpid() % 100), "-m", "npore_amd.realign", "--bam", bp, "--ref", fa,
                          "--out_prefix", str(tmp_path / "cli2")], cwd=REPO, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "indexed reader" not in out.stdout
    assert recs(str(tmp_path / "cli2.sam")) == recs(str(tmp_path / "cli1.sam")) == recs(str(whole)) and len(recs(str(whole))) == 260


def test_realign_cli_two_processes(tmp_path):
    """`torch.distributed.run --nproc-per-node 2 -m npore_amd.realign`: reads dealt by index over the ranks
    (both on this box's one GPU), part files merged by rank 0 -- same records as the reference's golden SAM."""
    import subprocess
    import sys
    from conftest import REPO
    prefix = str(tmp_path / "mp")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", str(29800 + os.getpid() % 100), "-m",
                           "npore_amd.realign", "--bam", os.path.join(GOLDEN, "data", "reads.bam"),
                           "--ref", os.path.join(GOLDEN, "data", "ref.fasta"), "--out_prefix", prefix], cwd=REPO)
    recs = lambda path: sorted(l for l in open(path) if not l.startswith("@"))
    got, want = recs(prefix + ".sam"), recs(os.path.join(GOLDEN, "data", "npore_realigned.sam"))
    assert len(got) == 10 and got == want
    assert not os.path.exists(prefix + ".part0.sam") and not os.path.exists(prefix + ".part1.sam")
    # the same with the BAM STREAMED (bounded-memory ingest): local rank 0 shares the record index, every rank takes
    # a contiguous share of the reads and inflates only the blocks that hold it
    env = dict(os.environ, NPORE_BAM_STREAM="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", str(29900 + os.getpid() % 100), "-m",
                           "npore_amd.realign", "--bam", os.path.join(GOLDEN, "data", "reads.bam"),
                           "--ref", os.path.join(GOLDEN, "data", "ref.fasta"), "--out_prefix", prefix + "_s"], cwd=REPO, env=env)
    assert recs(prefix + "_s.sam") == want


def test_long_polymers_and_many_periods(ctx, tables):
    """Engineered repeats: n-polymers longer than the LDS score table (L >= 32) and longer than max_l,
    units whose repeats are n-polymers for several periods at once (more than two SHR candidates in a
    column), with copy-number changes between read and reference -- the generic SHR pass, the global-memory
    score fallback and the "more periods" loop -- against the oracle at three band widths."""
    sub, nps = tables
    rng = np.random.default_rng(12)
    A, C, G, T = 1, 2, 3, 4
    blocks = [[A] * 150, [A, C] * 70, [A, C, G] * 45, [A] * 40 + [C] * 33, [A, A, C, A, A, C] * 30,
              [G, T, G, T, G, T, G, T, A] * 12, [T] * 101, [C, A, G, T] * 36, [A] * 12 + [A, C] * 9 + [A, C, G] * 7]
    refs, seqs, cigs = [], [], []
    for k in range(12):
        ref, seq, cig = [], [], []
        for b in rng.permutation(len(blocks))[:6]:
            unit = blocks[b]
            flank = [int(x) for x in rng.integers(1, 5, int(rng.integers(5, 30)))]
            ref += flank; seq += flank; cig += ["="] * len(flank)
            drop = int(rng.integers(0, 9)) * (1 if k % 2 else -1)      # copy-number change: read shorter / longer
            ref += unit
            if drop >= 0:
                seq += unit[:len(unit) - drop]; cig += ["="] * (len(unit) - drop) + ["D"] * drop
            else:
                seq += unit + unit[:(-drop)]; cig += ["="] * len(unit) + ["I"] * (-drop)
        refs.append(np.array(ref, np.uint8)); seqs.append(np.array(seq, np.uint8)); cigs.append("".join(cig))
    for r in (30, 100, 12):
        got, st = ctx.align_batch(refs, seqs, cigs, r=r, return_status=True)
        for k in range(len(refs)):
            want, wst = oracle.align(refs[k], seqs[k], cigs[k], sub, nps, r=r, return_status=True)
            assert got[k] == want and st[k] == wst, (r, k)


@pytest.mark.parametrize("r", [3, 12, 30, 31])
def test_narrow_band_window_over_long_deletions(ctx, tables, r):
    """Chunks of ONE wave keep a 128-entry reference-L window that is refilled 32 positions at a time
    (kernels.hpp WIN_STEP / WIN_SLACK): input paths with deletions of several hundred bases -- the band runs along
    the reference for many windows without a read step -- through repeat-rich sequence, so that LEN / generic SHR
    passes read the window all the way (including n-polymers that straddle a refill), against the oracle."""
    sub, nps = tables
    rng = np.random.default_rng(40 + r)
    A, C, G, T = 1, 2, 3, 4
    units = [[A] * 9, [A, C] * 6, [A, C, G] * 5, [T] * 40, [C, A] * 20, [G, G, T] * 11]
    refs, seqs, cigs = [], [], []
    for k in range(10):
        ref, seq, cig = [], [], []
        for piece in range(8):
            flank = [int(x) for x in rng.integers(1, 5, int(rng.integers(3, 40)))]
            ref += flank; seq += flank; cig += ["="] * len(flank)
            u = units[int(rng.integers(len(units)))]
            ref += u; seq += u[:len(u) - (piece % 3)]; cig += ["="] * (len(u) - piece % 3) + ["D"] * (piece % 3)
            if piece % 2 == 0:                      # a long deletion made of repeats and random bases
                gap = []
                while len(gap) < int(rng.integers(150, 700)):
                    gap += units[int(rng.integers(len(units)))] if rng.random() < 0.5 else [int(x) for x in rng.integers(1, 5, 17)]
                ref += gap; cig += ["D"] * len(gap)
            if piece == 5:                          # ... and a long insertion for the other direction
                ins = [int(x) for x in rng.integers(1, 5, 260)]
                seq += ins; cig += ["I"] * len(ins)
        refs.append(np.array(ref, np.uint8)); seqs.append(np.array(seq, np.uint8)); cigs.append("".join(cig))
    for mbr in (20000, 333):
        got, st = ctx.align_batch(refs, seqs, cigs, r=r, max_b_rows=mbr, return_status=True)
        for k in range(len(refs)):
            want, wst = oracle.align(refs[k], seqs[k], cigs[k], sub, nps, r=r, max_b_rows=mbr, return_status=True)
            assert got[k] == want and st[k] == wst, (r, mbr, k)


def test_traceback_drifting_paths_and_status(tables):
    """The traceback (one kernel since round 5: a 64-column group of the row landed on and of the row below per request)
    records the oracle's runs: strings and status bits equal at several band widths, incl. tiny max_b_rows (many chunks)
    and reads whose input CIGAR is far from the best path (the path drifts through the band's groups and leaves it)."""
    sub, nps = tables
    c = aln.Context(sub, nps, max_n=6, max_l=100, device=0)
    rng = np.random.default_rng(44)
    refs, seqs, cigs = synth.make_batch(808, 10, ref_len=1200, p_np=0.1)
    # a few reads with a deliberately bad input path: all insertions first, then all deletions, then matches
    for k in range(3):
        ref, seq = refs[k], seqs[k]
        m = min(len(ref), len(seq)) - 40
        refs.append(ref); seqs.append(seq)
        cigs.append("I" * (len(seq) - m) + "D" * (len(ref) - m) + "=" * m)
    # (the row kernel holds one 64-column group of an anti-diagonal: the drifting paths cross groups at r >= 64 -- at
    # r = 255 / 400 through several of them -- and leave the band at r = 10)
    for r, mbr in ((30, 20000), (100, 20000), (140, 300), (10, 37), (255, 500), (400, 150)):
        got, st = c.align_batch(refs, seqs, cigs, r=r, max_b_rows=mbr, return_status=True)
        for k in range(len(refs)):
            if r > 255 and k % 3:
                continue                            # (the oracle's state matrix at r = 400 is slow to set up: a third of the reads)
            want, wst = oracle.align(refs[k], seqs[k], cigs[k], sub, nps, r=r, max_b_rows=mbr, return_status=True)
            assert got[k] == want and st[k] == wst, (r, mbr, k)
    c.close()


def test_device_glue_equals_host_glue(ctx, tables):
    """npore_align_batch_cigars (realign_read's glue on the device: one lane per read runs csrc/std_stream.hpp over the
    traceback runs) == npore_align_batch followed by the host glue (npore_standardize_batch, the same header on the op
    string) and == the Python restatement on the oracle's strings: long reads, many chunks per read (tiny max_b_rows),
    low-complexity reads whose indel runs travel far and meet, a refused read (empty text, same status bits), reads whose
    input path is far from the best one (status bits, truncated strings)."""
    from npore_amd import cig
    sub, nps = tables
    refs, seqs, cigs = synth.make_batch(909, 24, ref_len=3000, p_np=0.2)
    r2, s2, c2 = synth.make_batch(910, 40, ref_len=400, p_np=0.6)
    refs += r2; seqs += s2; cigs += c2
    rng = np.random.default_rng(5)
    for k in range(12):                                      # two-letter reads: everything is a repeat
        n = int(rng.integers(50, 400))
        ref = rng.integers(1, 3, size=n).astype(np.uint8)
        ops, seq, j = [], [], 0
        while j < n:
            e = rng.random()
            if e < 0.12:
                ops.append("D"); j += 1
            elif e < 0.24:
                ops.append("I"); seq.append(int(rng.integers(1, 3)))
            else:
                ops.append("="); seq.append(int(ref[j])); j += 1
        refs.append(ref); seqs.append(np.array(seq, np.uint8)); cigs.append("".join(ops))
    for k in range(3):                                       # a bad input path
        ref, seq = refs[k], seqs[k]
        m = min(len(ref), len(seq)) - 40
        refs.append(ref); seqs.append(seq)
        cigs.append("I" * (len(seq) - m) + "D" * (len(ref) - m) + "=" * m)
    refs.append(refs[0]); seqs.append(seqs[0]); cigs.append(cigs[0][:-3])      # lengths disagree: refused
    for r, mbr in ((30, 20000), (100, 700), (10, 37)):
        raw, st = ctx.align_batch(refs, seqs, cigs, r=r, max_b_rows=mbr, return_status=True)
        fin, st2 = ctx.align_batch(refs, seqs, cigs, r=r, max_b_rows=mbr, return_status=True, final_cigars=True)
        assert (st == st2).all(), (r, mbr)
        assert st[-1] & 32 and fin[-1] == ""
        assert fin == cig.standardize_batch(raw, refs, seqs), (r, mbr)
        for k in (0, 30, 70, len(refs) - 3):
            want = oracle.align(refs[k], seqs[k], cigs[k], sub, nps, r=r, max_b_rows=mbr)
            assert fin[k] == cig.collapse_cigar(cig.standardize(want, refs[k], seqs[k])), (r, mbr, k)


@pytest.mark.parametrize("max_n,max_l", [(6, 20), (4, 20), (1, 5), (6, 31), (6, 32), (3, 127), (6, 5), (6, 3), (4, 2)])
def test_other_table_shapes(tables, max_n, max_l):
    """Contexts with other max_n / max_l (the CLI's --max_n / --max_l): row clamp at max_l - 1 also where the
    capped repeat count goes through the descriptor's table address (max_l < 32); max_l < max_n: periods above max_l
    score np_score's constant 100 (src/aln.pyx:265 with max_l in max_n's place)."""
    from test_model_vs_oracle import polymer_pairs, small_tables
    sub, nps = tables
    t = small_tables(nps, max_n, max_l, max_l)
    c = aln.Context(sub, t, max_n=max_n, max_l=max_l, device=0)
    pairs = polymer_pairs(100 + max_l, 25) + [(r_, s_, cg.decode()) for r_, s_, cg in zip(*synth.make_batch(55, 6, ref_len=1500, p_np=0.2))]
    for r in (5, 30, 64, 100):
        for mbr in (20000, 64):
            got, st = c.align_batch([p[0] for p in pairs], [p[1] for p in pairs], [p[2] for p in pairs], r=r, max_b_rows=mbr,
                                    return_status=True)
            for k, (ref, seq, cig) in enumerate(pairs):
                want, wst = oracle.align(ref, seq, cig, sub, t, max_b_rows=mbr, r=r, max_n=max_n, max_l=max_l, return_status=True)
                assert got[k] == want and st[k] == wst, (max_n, max_l, r, mbr, k)
    for seq in (pairs[0][0], pairs[3][1]):
        assert np.array_equal(c.get_np_info(seq), np.asarray(oracle.get_np_info(seq, max_n=max_n, max_l=max_l)))
    c.close()


def test_cli_max_l_with_shipped_table(tables, tmp_path):
    """--max_l 50 (reference CLI flag, src/realign.py:36-37): the reference hands align() the shipped
    [6,101,101] table whatever the flag says and clamps its indices at max_l - 1.  Context takes the full table
    and slices it; the strings equal the oracle's on the sliced table, and the CLI runs end to end."""
    import subprocess
    import sys
    from conftest import REPO
    sub, nps = tables
    c = aln.Context(sub, nps, max_n=6, max_l=50, device=0)
    refs, seqs, cigs = synth.make_batch(56, 6, ref_len=1500, p_np=0.2)
    got = c.align_batch(refs, seqs, cigs, r=30)
    t = np.ascontiguousarray(nps[:, :51, :51])
    for k in range(6):
        assert got[k] == oracle.align(refs[k], seqs[k], cigs[k], sub, t, r=30, max_l=50)
    c.close()
    prefix = str(tmp_path / "ml")
    subprocess.check_call([sys.executable, "-m", "npore_amd.realign", "--bam", os.path.join(GOLDEN, "data", "reads.bam"),
                           "--ref", os.path.join(GOLDEN, "data", "ref.fasta"), "--out_prefix", prefix,
                           "--max_l", "50", "--max_n", "4"], cwd=REPO)
    assert sum(1 for l in open(prefix + ".sam") if not l.startswith("@")) == 10


def test_calc_confusion_matrices_with_device_np_info():
    """calc_confusion_matrices (reference src/bam.pyx:351-499) with its n-polymer annotation from the GPU
    (aln.get_np_info on the range, as the reference calls its own) == the reference's matrices (tests/golden/cms.json)."""
    from npore_amd import bam, cfg
    import argparse
    g = load_json("cms.json")
    old = cfg.args
    cfg.args = argparse.Namespace(max_n=g["max_n"], max_l=g["max_l"])
    try:
        for c in g["cases"]:
            subs, nps, inss, dels = bam.calc_confusion_matrices((c["contig"], c["start"], c["end"]), pileups=c["lines"],
                                                                refs={c["contig"]: c["seq"]})
            want = np.zeros_like(nps)
            for a, b, d, v in c["nps_nonzero"]:
                want[a, b, d] = v
            assert subs.tolist() == c["subs"] and inss.tolist() == c["inss"] and dels.tolist() == c["dels"]
            assert np.array_equal(nps, want)
    finally:
        cfg.args = old


def test_realign_cli_recalc_cms(tmp_path, monkeypatch):
    """`realign --recalc_cms --recalc_exit` in the default native I/O mode (reference src/realign.py:81-95,
    src/bam.pyx:166-200) with get_pileups stubbed (samtools is absent): the matrices land in --stats_dir -- never in the
    package's data directory -- and equal calc_confusion_matrices on the same lines; without --stats_dir they go to
    ./stats like the reference."""
    import hashlib
    from npore_amd import bam, cfg, realign
    from npore_amd.bed import get_ranges
    shipped = os.path.join(os.path.dirname(os.path.abspath(bam.__file__)), "data", "guppy5_stats")
    digest = lambda: [hashlib.sha256(open(os.path.join(shipped, f), "rb").read()).hexdigest() for f in sorted(os.listdir(shipped))]
    before = digest()
    lines = {}

    def fake_pileups(bam_path, ctg, start, end):
        rng = np.random.default_rng(start + 7)
        out = []
        for _ in range(start, end):
            out.append("".join(rng.choice(list(".,ACGT*"), size=int(rng.integers(0, 6)))) + ("+2AC." if rng.random() < .05 else "")
                       + ("-1A," if rng.random() < .05 else ""))
        lines[(ctg, start, end)] = out
        return iter(out)

    monkeypatch.setattr(bam, "get_pileups", fake_pileups)
    monkeypatch.chdir(tmp_path)
    old = cfg.args
    try:
        for extra, d in ((["--stats_dir", str(tmp_path / "st")], tmp_path / "st"), ([], tmp_path / "stats")):
            cfg.args = realign.argparser().parse_args(
                ["--bam", os.path.join(GOLDEN, "data", "reads.bam"), "--ref", os.path.join(GOLDEN, "data", "ref.fasta"),
                 "--out_prefix", str(tmp_path / "o"), "--recalc_cms", "--recalc_exit", "--chunk_width", "20000"] + extra)
            with pytest.raises(SystemExit) as ex:
                realign.main()
            assert ex.value.code == 0
            got = [np.load(os.path.join(d, f"{k}_cm.npy")) for k in ("subs", "nps", "inss", "dels")]
            refs = bam.NativeFastaSeqs(os.path.join(GOLDEN, "data", "ref.fasta"))
            want = None
            for rg in get_ranges(cfg.args.regions, cfg.args.chunk_width):
                res = bam.calc_confusion_matrices(rg, pileups=lines[tuple(rg)], refs=refs)
                want = res if want is None else tuple(a + b for a, b in zip(want, res))
            assert want is not None and all(np.array_equal(a, b) for a, b in zip(got, want))
            assert got[0].sum() > 0
            assert not [f for f in os.listdir(d) if f.endswith(".tmp.npy")]
    finally:
        cfg.args = old
    assert digest() == before


def test_fill_shape_and_annotation_only_context(tables):
    """npore_fill_shape (launch geometry for reports) and a context without tables: get_np_info works,
    align is refused loudly."""
    sub, nps = tables
    c = aln.Context(None, None, max_n=6, max_l=100, device=0)
    sh = c.fill_shape(100)
    assert sh["waves_per_chunk"] == 4 and sh["resident_chunks"] == c.round_chunks(100)
    assert c.fill_shape(30)["waves_per_chunk"] == 1
    # single-wave chunks keep a 128-entry L window: 16 of them leave the CU the 4 KB the light kernels of the
    # neighbouring batches need (npore_api.cpp launch_fill), and so do the 4 four-wave chunks of r = 100
    for rr in (30, 100):
        sh = c.fill_shape(rr)
        assert sh["resident_waves_per_cu"] == 16 and sh["lds_bytes"] + 4096 <= 160 * 1024, (rr, sh)
    s = enc("ATATATATTTTTTAAAGCGCGC")
    assert np.array_equal(c.get_np_info(s), oracle.get_np_info(s))
    with pytest.raises(aln.NporeError, match="without penalty tables"):
        c.align_batch([s], [s], ["=" * len(s)])
    c.close()


@pytest.mark.gpu
def test_round_chunks(ctx):
    """npore_round_chunks: the batch-sizing hint of the C ABI (chunks the GPU holds at a time)."""
    assert ctx.round_chunks(30) > ctx.round_chunks(100) > ctx.round_chunks(200) > 0
    assert ctx.round_chunks(100) % 4 == 0          # four chunks share a workgroup at r = 100
    assert ctx.round_chunks(300) > 0 and ctx.round_chunks(511) > 0      # one chunk of 10 / 16 waves per workgroup
    assert ctx.round_chunks(512) == 0              # band wider than the kernels cover
