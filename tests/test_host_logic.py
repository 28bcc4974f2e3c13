"""CPU tests of the host side: C-ABI surface, glue, BAM reader, tables, sharding."""
import argparse
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import oracle
from npore_amd import _lib, aln, bam, cfg, cig, dist, synth
from conftest import load_json, GOLDEN, REPO


def test_header_symbols_exported_and_bound():
    """Every function include/npore_amd.h declares is exported by the built library and
    bound in npore_amd/_lib.py (no compute calls: there is no GPU here)."""
    hdr = open(os.path.join(REPO, "include", "npore_amd.h")).read()
    declared = set(re.findall(r"\b(npore_[a-z_0-9]+)\s*\(", hdr)) - {"npore_ctx"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.npore_abi_version() == 2
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (npore_[a-z_0-9]+)", out))
    assert declared <= exported


def test_no_gpu_fails_loudly(tables):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    sub, nps = tables
    with pytest.raises(aln.NporeError):
        aln.Context(sub, nps, max_n=6, max_l=100)      # no CPU fallback behind the C ABI


def test_product_does_not_touch_oracle():
    """Nothing under npore_amd/ imports, includes, links or executes oracle/ or tests/model."""
    py = re.compile(r"^\s*(import|from)\s+(oracle|tests|model)\b|oracle[./]|libnpore_oracle|libpull_model", re.M)
    cc = re.compile(r"#\s*include\s*[<\"][^>\"]*(oracle|tests/|host_prep|pull_model)")
    for root, _, files in os.walk(os.path.join(REPO, "npore_amd")):
        for f in files:
            txt = open(os.path.join(root, f), errors="replace").read() if f.endswith((".py", ".hpp", ".cpp", ".h")) else ""
            assert not (py.search(txt) if f.endswith(".py") else cc.search(txt)), f


def test_calc_score_matrices_golden(tables):
    """Row T: tables from the shipped count matrices == the reference's (G1), bit for bit."""
    sub, nps = tables
    cfg.args = argparse.Namespace(max_n=6, max_l=100)
    s, n, ins, dels = aln.load_default_tables()
    assert s.dtype == np.float32 and n.dtype == np.float32
    assert np.array_equal(s, sub) and np.array_equal(n, nps)
    z = np.load(os.path.join(GOLDEN, "tables.npz"))
    assert np.array_equal(ins, z["ins_scores"]) and np.array_equal(dels, z["del_scores"])


def test_cigar_glue_golden():
    """expand/collapse, bases_to_int and the one-pass standardisation reproduce the final
    CIGARs of the reference's test/data/npore_realigned.sam from its raw align() strings."""
    assert cig.expand_cigar("1D3M2I") == "DMMMII" and cig.collapse_cigar("DMMMII") == "1D3M2I"
    assert cig.bases_to_int("NACGT-").tolist() == [0, 1, 2, 3, 4, 5]
    fasta = bam.read_fasta(os.path.join(GOLDEN, "data", "ref.fasta"))["ref"]
    want = {r["name"]: r for r in load_json("reads_e2e.json")}
    golden = {}
    for line in open(os.path.join(GOLDEN, "data", "npore_realigned.sam")):
        if not line.startswith("@"):
            f = line.split("\t")
            golden[f[0]] = f[5]
    for line in open(os.path.join(GOLDEN, "data", "reads.sam")):
        if line.startswith("@"):
            continue
        f = line.rstrip("\n").split("\t")
        ex = cig.expand_cigar(f[5]).replace("S", "").replace("H", "")
        start = int(f[3]) - 1
        rlen = sum(1 for c in ex if c in "XD=M")
        final = cig.collapse_cigar(cig.standardize(want[f[0]]["raw_align"], cig.bases_to_int(fasta[start:start + rlen]),
                                                   cig.bases_to_int(f[9].upper())))
        assert final == want[f[0]]["final_cigar"] == golden[f[0]]


def test_cpp_glue_equals_python_glue(tables):
    """npore_standardize_batch (C++) == cig.standardize + collapse_cigar (Python restatement) on the
    golden reads and on oracle alignments of random reads (indels next to repeats)."""
    import oracle
    sub, nps = tables
    alns, refs, seqs = [], [], []
    for k in range(60):
        ref, seq, c = synth.make_pair(31, k, 200 + 13 * k, 0.15, 0.6)
        alns.append(oracle.align(ref, seq, c, sub, nps, r=10)); refs.append(ref); seqs.append(seq)
    alns += ["", "IIII", "DDDD", "=" * 7]
    refs += [np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.array([1, 1, 1, 1], np.uint8), np.array([1, 2, 3, 4, 1, 2, 3], np.uint8)]
    seqs += [np.zeros(0, np.uint8), np.array([2, 2, 2, 2], np.uint8), np.zeros(0, np.uint8), np.array([1, 2, 3, 4, 1, 2, 3], np.uint8)]
    want = [cig.collapse_cigar(cig.standardize(a, r_, s_)) for a, r_, s_ in zip(alns, refs, seqs)]
    assert cig.standardize_batch(alns, refs, seqs) == want
    assert cig.standardize_batch(alns, refs, seqs, threads=1) == want
    # the expanded form (what realign_hap returns), and expand_cigar's vectorised twin for long strings
    ops = [cig.standardize(a, r_, s_) for a, r_, s_ in zip(alns, refs, seqs)]
    assert cig.standardize_batch(alns, refs, seqs, expanded=True) == ops
    assert [cig._expand_cigar_np(w) for w in want] == ops == [cig.expand_cigar(w) for w in want]


def test_run_based_glue_equals_per_op_restatement():
    """The product's standardisation works on runs (npore_amd/cig.py, csrc/glue.hpp); oracle/glue_literal.py is the
    per-op restatement of the reference's push_indels_left / push_inss_thru_dels / one-pass glue (src/cig.pyx:102-192,
    src/bam.pyx:65-78).  Random edit scripts over low-complexity sequences (long periodic stretches, so that INDEL
    runs travel far and meet), runs at both ends, adjacent I / D blocks in every order: both product forms == literal."""
    from oracle import glue_literal
    rng = np.random.default_rng(11)
    alns, refs, seqs = [], [], []
    for case in range(400):
        alphabet = int(rng.integers(1, 4))                       # 1-3 distinct bases: everything is a repeat
        n_ops = int(rng.integers(0, 120))
        p = rng.dirichlet((2.0, 1.0, 1.0)) if case % 3 else np.array([0.2, 0.4, 0.4])
        ops = rng.choice(3, size=n_ops, p=p)                     # 0 M, 1 I, 2 D
        ref = rng.integers(1, alphabet + 1, size=int((ops != 1).sum())).astype(np.uint8)
        ins = rng.integers(1, alphabet + 1, size=int((ops == 1).sum())).astype(np.uint8)
        seq, i, j = [], 0, 0
        for op in ops:
            if op == 0:
                seq.append(ref[j]); j += 1
            elif op == 1:
                seq.append(ins[i]); i += 1
            else:
                j += 1
        aln = "".join("=ID"[o] if o else ("=" if rng.random() < 0.9 else "X") for o in ops)
        alns.append(aln); refs.append(ref); seqs.append(np.array(seq, np.uint8))
    want = [glue_literal.standardize(a, r_, s_) for a, r_, s_ in zip(alns, refs, seqs)]
    assert [cig.standardize(a, r_, s_) for a, r_, s_ in zip(alns, refs, seqs)] == want
    assert cig.standardize_batch(alns, refs, seqs, expanded=True) == want
    assert cig.standardize_batch(alns, refs, seqs) == [cig.collapse_cigar(w) for w in want]
    # the pieces on their own
    for a, r_, s_ in list(zip(alns, refs, seqs))[:100]:
        codes = [0 if c in "X=M" else (1 if c == "I" else 2) for c in a]
        lit = glue_literal.push_indels_left(list(codes), r_.tolist(), 2)
        assert cig.to_runs(lit) == cig.push_indels_left_runs(cig.to_runs(codes), r_.tolist(), 2)
        lit2 = glue_literal.push_inss_thru_dels(list(lit))
        assert cig.to_runs(lit2) == cig.inss_before_dels_runs(cig.to_runs(lit))


def test_bam_reader_equals_sam():
    """reads.bam decoded with zlib+struct == reads.sam field for field (pysam-free ingest)."""
    b = bam.BamFile(os.path.join(GOLDEN, "data", "reads.bam"))
    assert b.references == ["ref"] and b.lengths == [1001]
    refs = bam.read_fasta(os.path.join(GOLDEN, "data", "ref.fasta"))
    cfg.args = argparse.Namespace(bam="b", ref="r", contig=None, contigs=None, bed=None, contig_beg=None,
                                  contig_end=None, max_reads=0, max_n=6, max_l=100)
    assert bam.get_bam_regions(b, refs) == [("ref", 0, 1000)]
    rds = list(bam.get_read_data(b, refs))
    sam = {}
    for line in open(os.path.join(GOLDEN, "data", "reads.sam")):
        if not line.startswith("@"):
            f = line.rstrip("\n").split("\t")
            sam[f[0]] = f
    assert len(rds) == 10
    for rd in rds:
        f = sam[rd[0]]
        assert (int(f[1]), int(f[3]) - 1, int(f[4]), f[5], f[9], f[10], f[11]) == \
               (rd[1], rd[3], rd[4], rd[5], rd[7], rd[8], f"HP:i:{rd[10]}")
        assert len(rd[9]) == rd[6] - rd[3]
    cfg.args.max_reads = 3
    assert len(list(bam.get_read_data(b, refs))) == 3


def test_synth_deterministic():
    a = synth.make_pair(2, 5)
    b = synth.make_pair(2, 5)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    ops = np.frombuffer(a[2], np.uint8)
    assert np.isin(ops, [ord("="), ord("X"), ord("I")]).sum() == len(a[1])
    assert np.isin(ops, [ord("="), ord("X"), ord("D")]).sum() == len(a[0])


def _gloo_worker(rank, world_size, port, q):
    import torch.distributed as tdist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size))
    tdist.init_process_group("gloo", rank=rank, world_size=world_size)
    mine = list(dist.shard_indices(1001, rank, world_size))
    sums, maxes = dist.reduce_counters({"reads": len(mine), "cells": sum(mine)}, {"elapsed": 1.0 + rank})
    q.put((rank, mine[:3], sums, maxes))
    tdist.destroy_process_group()


def test_sharding_and_reduction_gloo_world2():
    """N>1 path on CPU: reads dealt round-robin, counters reduced with one sum + one max."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3, 5]
    for _, _, sums, maxes in res:
        assert sums["reads"] == 1001 and sums["cells"] == sum(range(1001)) and maxes["elapsed"] == 2.0


def _rank0_worker(rank, world_size, port, fail, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size),
                      LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world_size))
    calls = []

    def work():
        calls.append(rank)
        if fail:
            sys.exit(3)              # the way rank 0 leaves when samtools is missing (bam.get_pileups)
        return "done"
    try:
        q.put((rank, dist.rank0_then_all(work), calls, dist.host_threads_per_rank()))
    except SystemExit as e:
        q.put((rank, f"exit {e.code}", calls, 0))
    except RuntimeError as e:
        q.put((rank, f"error {e}", calls, 0))


def test_rank0_then_all_gloo_world2():
    """dist.rank0_then_all (the host-only steps of a multi-process run, e.g. --recalc_cms): rank 0 works, every rank
    learns the outcome -- also when rank 0 leaves through sys.exit: the others raise instead of waiting at a barrier
    for the process group's timeout.  host_threads_per_rank divides what the cgroup quota leaves by the local ranks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    for fail in (False, True):
        q = ctx.Queue()
        port = 31500 + os.getpid() % 2000 + (7 if fail else 0)
        procs = [ctx.Process(target=_rank0_worker, args=(r, 2, port, fail, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = dict((r[0], r[1:]) for r in (q.get(timeout=120) for _ in procs))
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        assert res[0][1] == [0] and res[1][1] == []          # only rank 0 worked
        if fail:
            assert res[0][0] == "exit 3" and res[1][0].startswith("error rank 0 failed")
        else:
            assert res[0][0] == "done" and res[1][0] is None
            quota = dist.cgroup_cpus()
            cores = len(os.sched_getaffinity(0))
            want = max(1, (min(cores, max(1, int(quota + 0.5))) if quota is not None else cores) // 2)
            assert res[0][2] == res[1][2] == want


def test_bench_cpu_baseline_leg(tables):
    """bench.py's CPU-baseline leg on a tiny batch (no GPU involved): one core, a pool sweep, the k-scaled
    Cython-equivalent figures, and the strings it hands back for the comparison with the GPU output."""
    import argparse
    import bench
    sub, nps = tables
    refs, seqs, cigs = synth.make_batch(2, 6, ref_len=400)
    args = argparse.Namespace(cpu_sample=3, cpu_threads=0, max_b_rows=300, r=10)
    cpu, want = bench.cpu_baseline(args, refs, seqs, cigs, sub, nps)
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["value"] > 0
    assert set(want) == set(range(6))
    for k in range(6):
        assert want[k] == oracle.align(refs[k], seqs[k], cigs[k], sub, nps, max_b_rows=300, r=10)
    assert cpu["all_reads"]["cores"] >= 1 and cpu["all_reads"]["sweep"]
    kk = cpu["k_cython_over_port"]["k"]
    assert 0.2 < kk < 0.6                       # tests/golden/k_cython_over_port.json (measure_k.py)
    assert abs(cpu["cython_equivalent"]["value"] - cpu["value"] * kk) < 1e-2 * cpu["value"]
    hc = bench.host_cpus()
    assert 1 <= hc["usable"] <= hc["cpu_count"]
    assert len(bench.csrc_sha()) == 16


def test_bench_digest_parity_every_rank(tables):
    """bench.py's per-rank string check (any world size): rank k of N holds reads k, k + N, ... and compares each
    output whose read index has a committed digest of the pinned oracle; a flipped byte or a wrong length is counted
    as bad, reads without a digest and configurations without a table are skipped."""
    import bench
    sub, nps = tables
    z = np.load(os.path.join(REPO, "tests", "golden", "fullsize_digests.npz"))
    assert set(z["c2_idx"][:8000].tolist()) >= set(range(8000))          # what 8 ranks x 1 000 reads hold
    for world_size in (1, 2, 4, 8):                                     # every rank of every world size holds sampled reads
        for name in ("r30", "c4"):
            assert len(set((z[name + "_idx"] % world_size).tolist())) == world_size, (name, world_size)
    key = (2, False, 10_000, 100, 20000)
    world_size, n = 4, 3
    for rank in (0, 3):
        idx = [rank + k * world_size for k in range(n)]
        outs = [oracle.align(*synth.make_pair(2, i, 10_000), sub, nps, r=100).encode() for i in idx]
        oo = np.zeros(n + 1, np.int64)
        oo[1:] = np.cumsum([len(o) + 100 for o in outs])                 # slots larger than the strings, as in bench.py
        buf = np.zeros(int(oo[-1]), np.uint8)
        for k, o in enumerate(outs):
            buf[oo[k]:oo[k] + len(o)] = np.frombuffer(o, np.uint8)
        ln = np.array([len(o) for o in outs], np.int64)
        assert bench.digest_parity(key, idx, buf, oo, ln) == (n, 0)
        bad = buf.copy()
        bad[oo[1] + 5] ^= 1
        assert bench.digest_parity(key, idx, bad, oo, ln) == (n, 1)
        ln2 = ln.copy(); ln2[2] -= 1
        assert bench.digest_parity(key, idx, buf, oo, ln2) == (n, 1)
        assert bench.digest_parity(key, [10_000_000 + i for i in idx], buf, oo, ln) == (0, 0)      # no digest: skipped
        assert bench.digest_parity((9, False, 10_000, 100, 20000), idx, buf, oo, ln) == (0, 0)     # no table


def _parts_worker(rank, world_size, port, prefix, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size),
                      LOCAL_RANK=str(rank))
    with open(f"{prefix}.part{rank}.sam", "w") as fh:
        for k in range(rank, 7, world_size):
            fh.write(f"read{k}\n")
    q.put((rank, dist.gather_parts(prefix + ".sam", prefix, len(range(rank, 7, world_size)))))


def test_realign_part_files_gloo_world2(tmp_path):
    """The multi-process end of realign.py: each rank wrote its reads (rank, rank+N, ...) to a part file;
    one sum over gloo is the barrier, rank 0 appends the parts to the SAM and removes them."""
    import torch.multiprocessing as mp
    prefix = str(tmp_path / "o")
    open(prefix + ".sam", "w").write("@HD\n")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_parts_worker, args=(r, 2, port, prefix, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, 7), (1, 7)]
    assert open(prefix + ".sam").read().split() == ["@HD", "read0", "read2", "read4", "read6", "read1", "read3", "read5"]
    assert not os.path.exists(prefix + ".part0.sam") and not os.path.exists(prefix + ".part1.sam")


# ---- native host I/O (csrc/hostio.hpp) against the pure-Python restatement in npore_amd/bam.py ----
def _native_vs_python(bam_path, fasta_path, regions, max_reads=0, stream=None):
    import argparse
    from npore_amd import bam, cfg
    from npore_amd.cig import bases_to_int, expand_cigar
    old = cfg.args
    cfg.args = argparse.Namespace(max_n=6, max_l=100, regions=regions, max_reads=max_reads)
    try:
        py = bam.BamFile(bam_path)
        refs = bam.read_fasta(fasta_path)
        rds = list(bam.get_read_data(py, refs))
        nb, nf = bam.NativeBam(bam_path, threads=3, stream=stream), bam.NativeFasta(fasta_path)
        assert stream is None or nb.streamed == stream
        assert nb.references == py.references and nb.lengths == py.lengths
        assert nb.n_records == len(py.records) and nb.refs_with_reads() == py.refs_with_reads()
        assert list(nf) == list(refs) and all(len(nf[k]) == len(refs[k]) for k in refs)
        idx = nb.select(regions, max_reads)
        assert len(idx) == len(rds)
        assert [py.records[i].query_name for i in idx] == [rd[0] for rd in rds]
        r, ro, s, so, c, co = nb.pack(nf, idx, threads=3)
        for k, rd in enumerate(rds):
            assert np.array_equal(r[ro[k]:ro[k + 1]], bases_to_int(rd[9])), k
            assert np.array_equal(s[so[k]:so[k + 1]], bases_to_int(rd[7])), k
            assert c[co[k]:co[k + 1]].tobytes().decode() == expand_cigar(rd[5]).replace("S", "").replace("H", ""), k
        finals = [f"{7 + k}M{k % 3 + 1}I2D" for k in range(len(rds))]
        status = np.array([32 if k % 5 == 4 else (4 if k % 7 == 3 else 0) for k in range(len(rds))], np.int32)
        want = "".join(bam.sam_line(rd, f) for rd, f, st in zip(rds, finals, status) if not st & 32)
        assert nb.format_sam(idx, finals, status, threads=3) == want
        nb.close(); nf.close()
        return len(rds)
    finally:
        cfg.args = old


def test_native_bam_matches_python_reader_on_reference_data():
    d = os.path.join(GOLDEN, "data")
    assert _native_vs_python(os.path.join(d, "reads.bam"), os.path.join(d, "ref.fasta"), [("ref", 0, 1000)]) == 10
    assert _native_vs_python(os.path.join(d, "reads.bam"), os.path.join(d, "ref.fasta"), [("ref", 0, 1000)], max_reads=3) == 3
    assert _native_vs_python(os.path.join(d, "reads.bam"), os.path.join(d, "ref.fasta"), [("ref", 400, 450), ("ref", 0, 30)]) > 0


def test_calc_confusion_matrices_golden():
    """calc_confusion_matrices (reference src/bam.pyx:351-499) against tests/golden/cms.json: the count matrices the
    reference's compiled function returns for the same pileup lines (made by tests/golden/make_golden_cms.py from the
    reference's test reads, plus engineered lines: '^' + odd mapping qualities, '$', '*', lower case, n-polymer and
    other INDELs, lengths beyond max_l, an unexpected character).  get_np_info comes from the oracle here (no GPU)."""
    g = load_json("cms.json")
    old = cfg.args
    cfg.args = argparse.Namespace(max_n=g["max_n"], max_l=g["max_l"])
    try:
        for c in g["cases"]:
            seq, start, end = c["seq"], c["start"], c["end"]
            info = oracle.get_np_info(cig.bases_to_int(seq[start:end + 1]))
            for threads in (1, 3):
                subs, nps, inss, dels = bam.calc_confusion_matrices((c["contig"], start, end), pileups=c["lines"],
                                                                    refs={c["contig"]: seq}, np_info=info, threads=threads)
                assert subs.tolist() == c["subs"] and inss.tolist() == c["inss"] and dels.tolist() == c["dels"], (c["contig"], start)
                want = np.zeros_like(nps)
                for a, b, d, v in c["nps_nonzero"]:
                    want[a, b, d] = v
                assert np.array_equal(nps, want), (c["contig"], start, end)
        # ranges accumulate like the reference's pool results (src/bam.pyx:183-188): counting adds into the matrices
        c = g["cases"][1]
        info = oracle.get_np_info(cig.bases_to_int(c["seq"][c["start"]:c["end"] + 1]))
        a = bam.calc_confusion_matrices((c["contig"], c["start"], c["end"]), pileups=c["lines"], refs={c["contig"]: c["seq"]}, np_info=info)
        assert int(a[0].sum()) == int(np.array(c["subs"]).sum())
    finally:
        cfg.args = old


def test_inflated_bam_copy_shared_between_local_ranks(tmp_path, monkeypatch):
    """Several processes per node: local rank 0 inflates the BAM once and leaves the stream under /dev/shm
    (npore_bam_dump_inflated), the other local ranks map that copy (npore_bam_open recognises an inflated stream);
    same records either way; the maker removes the copy when it closes.  Part files are appended in the kernel."""
    from npore_amd import bam, dist
    path = os.path.join(GOLDEN, "data", "reads.bam")
    lib = _lib.load()
    plain = bam.NativeBam(path, share=False)
    raw = str(tmp_path / "copy.raw")
    assert lib.npore_bam_dump_inflated(plain.handle, raw.encode()) == 0
    assert os.path.getsize(raw) == lib.npore_bam_inflated_size(plain.handle) and open(raw, "rb").read(4) == b"BAM\1"
    mapped = bam.NativeBam(raw, share=False)
    regions = [("ref", 0, 1000)]
    assert mapped.references == plain.references and mapped.n_records == plain.n_records
    assert np.array_equal(mapped.select(regions), plain.select(regions))
    mapped.close()
    if os.path.isdir("/dev/shm"):
        monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
        monkeypatch.setenv("MASTER_PORT", str(40000 + os.getpid() % 20000))
        monkeypatch.setenv("LOCAL_RANK", "0")
        maker = bam.NativeBam(path)
        assert maker._shared and os.path.exists(maker._shared[0])
        monkeypatch.setenv("LOCAL_RANK", "1")
        user = bam.NativeBam(path)                  # finds the copy at once
        assert user._shared is None and np.array_equal(user.select(regions), plain.select(regions))
        shared = maker._shared[0]
        user.close(); maker.close()
        assert not os.path.exists(shared) and not os.path.exists(shared.replace(".raw", ".pid"))
        # the next open of the same file in the same run is a new generation: the `.skip` the first maker left for late
        # comers is not its successor's (a rank that is faster than rank 0 must WAIT for the new copy, not give up)
        assert os.path.exists(shared.replace(".raw", ".skip"))
        monkeypatch.setenv("LOCAL_RANK", "0")
        maker2 = bam.NativeBam(path)
        assert maker2._shared and maker2._shared[0] != shared and os.path.exists(maker2._shared[0])
        monkeypatch.setenv("LOCAL_RANK", "1")
        user2 = bam.NativeBam(path)
        assert np.array_equal(user2.select(regions), plain.select(regions))
        user2.close(); maker2.close()
        # a maker that died before it left anything: the waiting rank notices and opens the file itself
        key = "deadbeefdeadbeef"
        import subprocess
        pr = subprocess.Popen([sys.executable, "-c", "pass"]); pr.wait()
        open(f"/dev/shm/npore_bam_{key}.pid", "w").write(str(pr.pid))
        assert bam.NativeBam._wait_for_maker(key, f"/dev/shm/npore_bam_{key}.raw", f"/dev/shm/npore_bam_{key}.skip", 30.0) is False
        os.remove(f"/dev/shm/npore_bam_{key}.pid")
        # a maker that never shows up (no pid file: the ranks' keys differ, or local rank 0 does not run this code): the
        # waiting rank gives up after the grace period, long before the sharing timeout
        import time
        monkeypatch.setenv("NPORE_SHARE_GRACE_S", "0.3")
        t0 = time.time()
        assert bam.NativeBam._wait_for_maker("feedfacefeedface", "/dev/shm/npore_bam_feedfacefeedface.raw",
                                             "/dev/shm/npore_bam_feedfacefeedface.skip", 60.0) is False
        assert time.time() - t0 < 5.0
        # the key holds nothing per-process: a rank started by another parent (a per-rank wrapper script) computes the same
        code = ("import sys; sys.path.insert(0, %r); from npore_amd import bam; print(bam.NativeBam._shm_key(%r))" % (REPO, path))
        monkeypatch.setenv("LOCAL_RANK", "1")
        k_direct = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).stdout.strip()
        k_wrapped = subprocess.run(["bash", "-c", f"true; {sys.executable} -c {code!r}; true"], capture_output=True, text=True).stdout.strip()
        assert len(k_direct) == 16 and k_direct == k_wrapped
        for f in (shared.replace(".raw", ".skip"), maker2._shared and "" or ""):
            if f and os.path.exists(f):
                os.remove(f)
    plain.close()
    # dist._append_file: whole files, also behind existing content
    a, b = tmp_path / "a", tmp_path / "b"
    a.write_bytes(b"@HD\n"); b.write_bytes(b"x" * 100_000 + b"\n")
    with open(a, "ab") as out, open(b, "rb") as src:
        dist._append_file(out, src)
        out.write(b"tail\n")
    assert a.read_bytes() == b"@HD\n" + b"x" * 100_000 + b"\ntail\n"
    assert dist.host_threads_per_rank() >= 1


def test_inflate_decoder_equals_zlib(tmp_path):
    """csrc/inflate.hpp (the BGZF readers' DEFLATE decoder) against zlib: every block type (stored, fixed, dynamic), every
    compression level and strategy, data from incompressible to one long run, the blocks of the test BAMs; streams it
    declines (a single-symbol literal code) still inflate through the zlib fallback; corrupted and truncated streams are
    refused or inflate to something -- never read or written out of bounds (the output buffer is guarded)."""
    import zlib
    lib = _lib.load()
    rng = np.random.default_rng(11)

    def inflate(raw, n, force):
        out = np.full(n + 64, 0xA5, np.uint8)                      # 32 guard bytes either side
        src = np.frombuffer(raw, np.uint8)
        rc = lib.npore_debug_inflate(src.ctypes.data, len(raw), out[32:].ctypes.data, n, force)
        assert (out[:32] == 0xA5).all() and (out[32 + n:] == 0xA5).all(), "wrote outside the block"
        return rc, out[32:32 + n].tobytes()

    def deflate(data, level, strategy=zlib.Z_DEFAULT_STRATEGY):
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
        return c.compress(data) + c.flush()

    datas = [b"", b"A", b"ACGT" * 5000, bytes(rng.integers(0, 256, 60000, dtype=np.uint8)), bytes(rng.integers(0, 4, 65280, dtype=np.uint8)),
             b"\0" * 65280, bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), 40000)) + bytes(rng.integers(33, 75, 25000, dtype=np.uint8)),
             bytes(np.repeat(rng.integers(0, 256, 300, dtype=np.uint8), rng.integers(1, 300, 300)))[:65000]]
    def inflate_pair(raw_a, n_a, raw_b, n_b, force):
        oa, ob = np.full(n_a + 64, 0xA5, np.uint8), np.full(n_b + 64, 0x5A, np.uint8)
        sa, sb = np.frombuffer(raw_a, np.uint8), np.frombuffer(raw_b, np.uint8)
        rc = lib.npore_debug_inflate_pair(sa.ctypes.data, len(raw_a), oa[32:].ctypes.data, n_a, sb.ctypes.data, len(raw_b), ob[32:].ctypes.data, n_b, force)
        assert (oa[:32] == 0xA5).all() and (oa[32 + n_a:] == 0xA5).all() and (ob[:32] == 0x5A).all() and (ob[32 + n_b:] == 0x5A).all(), "wrote outside a block"
        return rc, oa[32:32 + n_a].tobytes(), ob[32:32 + n_b].tobytes()

    n_fast = n_all = 0
    streams = []                                                    # (raw, data) of every case, for the pairs below
    for data in datas:
        for level in (0, 1, 4, 6, 9):
            for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY, zlib.Z_FILTERED):
                raw = deflate(data, level, strat)
                rc2, got2 = inflate(raw, len(data), 2)
                assert rc2 == 1 and got2 == data
                rc0, got0 = inflate(raw, len(data), 0)
                assert rc0 == 1 and got0 == data, (len(data), level, strat)
                rc1, got1 = inflate(raw, len(data), 1)
                assert rc1 == 0 or got1 == data, (len(data), level, strat)
                n_fast += rc1; n_all += 1
                streams.append((raw, data, rc1))
                if len(data) > 1:                                   # a wrong announced size is refused by both
                    assert inflate(raw, len(data) - 1, 1)[0] == 0 and inflate(raw, len(data) + 1, 0)[0] == 0
    assert n_fast >= 0.9 * n_all, (n_fast, n_all)                   # the decoder takes nearly everything itself
    # two streams side by side (how the readers take a file's blocks): every case with a partner drawn at random -- long
    # with short, stored with dynamic, many deflate blocks with one -- gives what each gives alone, decoder only and with
    # the zlib fallback
    order = rng.permutation(len(streams))
    for i, j in zip(range(len(streams)), order.tolist()):
        (ra, da, fa), (rb, db, fb) = streams[i], streams[j]
        rc, ga, gb = inflate_pair(ra, len(da), rb, len(db), 0)
        assert rc == 3 and ga == da and gb == db, (i, j)
        rc, ga, gb = inflate_pair(ra, len(da), rb, len(db), 1)
        assert (rc & 1) == fa and (rc >> 1) == fb and (not fa or ga == da) and (not fb or gb == db), (i, j, rc)
    # the blocks of a real BAM
    for name in ("reads.bam",):
        raw = open(os.path.join(GOLDEN, "data", name), "rb").read()
        p, nb = 0, 0
        while p < len(raw):
            xlen = int.from_bytes(raw[p + 10:p + 12], "little")
            bsize = int.from_bytes(raw[p + 16:p + 18], "little") + 1
            isize = int.from_bytes(raw[p + bsize - 4:p + bsize], "little")
            pay = raw[p + 12 + xlen:p + bsize - 8]
            rc1, got1 = inflate(pay, isize, 1)
            assert rc1 == 1 and got1 == zlib.decompress(pay, -15)
            p += bsize; nb += 1
        assert nb >= 2
    # corrupted / truncated streams: no out-of-bounds access, and whatever is accepted has the announced size
    base = deflate(datas[6], 6)
    for k in range(300):
        bad = bytearray(base)
        for _ in range(int(rng.integers(1, 4))):
            bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        if k % 3 == 0:
            bad = bad[:int(rng.integers(1, len(bad)))]
        inflate(bytes(bad), len(datas[6]), 1)
        inflate(bytes(bad), len(datas[6]), 0)
        # beside a corrupted partner (either side) a good stream still inflates, and nobody writes outside its own buffer
        good_raw, good = streams[int(rng.integers(0, len(streams)))][:2]
        if k % 2:
            rc, _, gb = inflate_pair(bytes(bad), len(datas[6]), good_raw, len(good), 0)
            assert rc & 2 and gb == good
        else:
            rc, ga, _ = inflate_pair(good_raw, len(good), bytes(bad), len(datas[6]), 0)
            assert rc & 1 and ga == good


def test_one_pass_handle_reads_only_the_header():
    """npore_bam_open_mode(..., 3): block table + BAM header, no record index (the reads go through
    npore_bam_realign_sequential in one pass); a file that is not BGZF is refused."""
    path = os.path.join(GOLDEN, "data", "reads.bam")
    full, head = bam.NativeBam(path, share=False), bam.NativeBam(path, one_pass=True)
    assert head.one_pass and head.streamed and head.references == full.references and head.lengths == full.lengths
    assert head.n_records == 0 and len(head.select([("ref", 0, 1000)])) == 0 and head.refs_with_reads() == {0}
    lib = _lib.load()
    assert not lib.npore_bam_open_mode(os.path.join(GOLDEN, "data", "ref.fasta").encode(), 0, 3, None)
    assert "BGZF" in _lib.last_error()
    full.close(); head.close()


def test_native_bam_clips_flags_tags(tmp_path):
    """A synthetic BAM with soft/hard clips, IUPAC bases, missing qualities, HP tags of several widths,
    secondary / supplementary / unmapped records, two contigs and reads hanging over a contig end."""
    from npore_amd import bam
    rng = np.random.default_rng(11)
    contigs = {"chrA": "".join(rng.choice(list("ACGT"), 700)), "chrB": "".join(rng.choice(list("acgtN"), 300))}
    fa = tmp_path / "r.fa"
    fa.write_text("".join(f">{n} description\n" + "\n".join(s[i:i + 60] for i in range(0, len(s), 60)) + "\n" for n, s in contigs.items()))
    recs = []
    for k in range(60):
        rid = int(rng.integers(0, 2))
        clen = len(contigs["chrA" if rid == 0 else "chrB"])
        core = [(int(rng.choice([0, 7, 8, 1, 2])), int(rng.integers(1, 9))) for _ in range(int(rng.integers(1, 8)))]
        cig = list(core)
        if k % 3 == 0: cig = [(4, int(rng.integers(1, 6)))] + cig
        if k % 4 == 0: cig = cig + [(4, int(rng.integers(1, 6)))]
        if k % 6 == 0: cig = [(5, 3)] + cig
        if k % 10 == 0: cig = cig + [(5, 2)]
        qlen = sum(n for op, n in cig if op in (0, 1, 4, 7, 8))
        rlen = sum(n for op, n in cig if op in (0, 2, 3, 7, 8))
        pos = int(rng.integers(0, max(1, clen - rlen + (3 if k % 9 == 0 else 0))))
        seq = "".join(rng.choice(list("ACGTNRY="), qlen, p=[.23, .23, .23, .23, .03, .02, .02, .01]))
        flag = [0, 16, 0x100, 0x800, 4, 0, 16, 0][k % 8]
        recs.append(dict(name=f"read{k}", flag=flag, ref_id=rid, pos=pos, mapq=int(rng.integers(0, 61)), cigar=cig, seq=seq,
                         qual=None if k % 5 == 0 else rng.integers(0, 42, qlen).astype(np.uint8).tobytes(),
                         hp=None if k % 3 == 1 else int(rng.integers(0, 3))))
    recs.sort(key=lambda r: (r["ref_id"], r["pos"]))
    bp = tmp_path / "x.bam"
    bam.write_bam(str(bp), [(n, len(s)) for n, s in contigs.items()], recs)
    n1 = _native_vs_python(str(bp), str(fa), [("chrA", 0, 699), ("chrB", 0, 299)])
    n2 = _native_vs_python(str(bp), str(fa), [("chrB", 100, 200)])
    n3 = _native_vs_python(str(bp), str(fa), [("chrA", 0, 699), ("chrB", 0, 299)], max_reads=7)
    assert n1 > 20 and 0 < n2 < n1 and n3 == 7
    # not coordinate-sorted: the per-reference index falls back to a scan in file order
    rng.shuffle(recs)
    bam.write_bam(str(tmp_path / "u.bam"), [(n, len(s)) for n, s in contigs.items()], recs)
    assert _native_vs_python(str(tmp_path / "u.bam"), str(fa), [("chrA", 0, 699), ("chrB", 0, 299)]) == n1
    assert _native_vs_python(str(tmp_path / "u.bam"), str(fa), [("chrB", 100, 200)]) == n2


def test_native_bam_rejects_corrupt_files(tmp_path):
    """Truncated / damaged BAM files are refused with a message (the reference: pysam raises), never walked."""
    from npore_amd import _lib, bam
    import zlib, struct
    lib = _lib.load()
    good = open(os.path.join(GOLDEN, "data", "reads.bam"), "rb").read()
    raw = bam._bgzf_decompress(os.path.join(GOLDEN, "data", "reads.bam"))

    def bgzf(payload, extra=b"", crc_of=lambda chunk: zlib.crc32(chunk)):
        out = b""
        for p in range(0, len(payload), 0xFF00):
            chunk = payload[p:p + 0xFF00]
            c = zlib.compressobj(6, zlib.DEFLATED, -15)
            data = c.compress(chunk) + c.flush()
            out += struct.pack("<BBBBIBBH", 31, 139, 8, 4, 0, 0, 0xFF, 6 + len(extra)) + extra + \
                struct.pack("<BBHH", 66, 67, 2, len(data) + 25 + len(extra)) + data + \
                struct.pack("<II", crc_of(chunk), len(chunk))
        return out

    # find the first record and damage its l_seq / cut the stream inside a record
    l_text, = struct.unpack_from("<i", raw, 4)
    p = 8 + l_text
    n_ref, = struct.unpack_from("<i", raw, p); p += 4
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", raw, p); p += 8 + l_name
    bad_lseq = bytearray(raw); struct.pack_into("<i", bad_lseq, p + 4 + 16, 1 << 28)
    cases = {"cut.bam": good[:len(good) // 2], "notbam.bam": b"hello world" * 10, "lseq.bam": bgzf(bytes(bad_lseq)),
             "short.bam": bgzf(raw[:p + 40]), "ok.bam": bgzf(raw),
             # the block table hops from header to header: a header longer than its first read (another extra subfield
             # of 100 bytes before 'BC'), a last member cut three bytes short, bytes after the last member
             "ok_longextra.bam": bgzf(raw, struct.pack("<BBH", 88, 89, 100) + bytes(100)),
             "cut3.bam": bgzf(raw)[:-3], "tail.bam": bgzf(raw) + b"\x00" * 40,
             # a member whose bytes inflate to the announced length but are not the bytes its CRC-32 was made of (htslib, the
             # reference's reader, raises there too)
             "crc.bam": bgzf(raw, crc_of=lambda chunk: zlib.crc32(chunk) ^ 0x10)}
    for name, content in cases.items():
        path = tmp_path / name
        path.write_bytes(content)
        h = lib.npore_bam_open(os.fsencode(str(path)), 2)
        if name.startswith("ok"):
            assert h and lib.npore_bam_n_records(h) == 10
            lib.npore_bam_close(h)
        else:
            assert not h and _lib.last_error(), name
            if name == "crc.bam":                     # every ingest mode inflates through the same check
                for mode in (1, 2):
                    assert not lib.npore_bam_open_mode(os.fsencode(str(path)), 2, mode, None) and "BGZF" in _lib.last_error(), mode


def test_crc32_equals_zlib():
    """csrc/crc32.hpp (folding by carry-less multiplication; zlib's tables for the bytes off the 16-byte grid and on CPUs
    without PCLMULQDQ) == zlib.crc32 on random lengths around the folding widths, unaligned starts and running values."""
    import zlib
    from npore_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(11)
    for t in range(1500):
        n = int(rng.choice([0, 1, 15, 16, 63, 64, 65, 79, 80, 127, 128, 129, 1000, 4096, 65280, 100000])) + int(rng.integers(0, 3))
        off = int(rng.integers(0, 17))
        a = rng.integers(0, 256, n + off, dtype=np.uint8)
        crc0 = int(rng.integers(0, 2 ** 32)) if t % 2 else 0
        assert lib.npore_debug_crc32(a.ctypes.data + off, n, crc0) == zlib.crc32(a[off:].tobytes(), crc0), (n, off, crc0)


def test_native_fasta_equals_python_reader(tmp_path):
    """The library's parallel FASTA parser (pieces cut at line starts) against read_fasta: odd line ends, blanks,
    lower case, empty contigs, a contig spanning several pieces, bases before the first header."""
    from npore_amd import bam
    rng = np.random.default_rng(8)
    big = "".join(rng.choice(list("ACGTacgtN"), 1_500_000))
    parts = ["ACGT\n", ">c1 description here\n", "acgtn\r\n", "\n", "  GG TT\t\n", ">empty\n", ">c3\tx\n"]
    parts += [big[i:i + 70] + ("\r\n" if (i // 70) % 1000 == 0 else "\n") for i in range(0, len(big), 70)]
    parts += [">c4\n", "TTTT", ]                       # no newline at the end
    path = tmp_path / "odd.fa"
    path.write_text("".join(parts))
    want = bam.read_fasta(str(path))
    nf = bam.NativeFasta(str(path))
    try:
        assert nf.names == list(want) == ["c1", "empty", "c3", "c4"]
        for name in nf.names:
            assert len(nf[name]) == len(want[name]) and nf.sequence(name) == want[name], name
        assert want["c1"] == "ACGTNGG TT" and len(want["c3"]) == 1_500_000
    finally:
        nf.close()
    lazy = bam.NativeFastaSeqs(str(path))
    assert list(lazy) == list(want) and len(lazy) == 4 and "c3" in lazy and "zz" not in lazy
    assert dict(lazy.items()) == want
    (tmp_path / "none.fa").write_text("")
    nf = bam.NativeFasta(str(tmp_path / "none.fa"))
    assert nf.names == []
    nf.close()
    with pytest.raises(SystemExit):
        bam.NativeFasta(str(tmp_path / "missing.fa"))


def test_experiment_builds_still_compile(tmp_path):
    """The measurement hooks DESIGN.md section 6 quotes (pad_hooks.hpp, -DNPORE_STATS, the poll sleep) are compiled out of
    the product; this keeps them compiling (device code only, gfx950)."""
    import shutil
    import subprocess
    if not shutil.which("hipcc"):
        pytest.skip("no hipcc")
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "npore_amd", "csrc", "npore_api.cpp")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-ffp-contract=off", "-x", "hip", "--cuda-device-only",
           "-DNPORE_STATS", "-DNPORE_PAD_VALU=4", "-DNPORE_PAD_SALU=4", "-DNPORE_PAD_NOP=2", "-DNPORE_PAD_OP=3",
           "-DNPORE_X_POLLSLEEP=1", "-c", "-o", str(tmp_path / "exp.o"), src]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]


def test_append_file_copy_file_range(tmp_path):
    """dist._append_file appends inside the kernel (copy_file_range needs a destination that is NOT O_APPEND) and
    leaves the position at the end, also through the read/write fallback."""
    from npore_amd import dist
    a, b, c = tmp_path / "a", tmp_path / "b", tmp_path / "c"
    a.write_bytes(b"header\n")
    b.write_bytes(b"x" * 100_000)
    c.write_bytes(b"tail")
    with open(a, "r+b") as out:
        out.seek(0, os.SEEK_END)
        calls = []
        real = getattr(os, "copy_file_range", None)
        if real is not None:
            def spy(*args, **kw):
                k = real(*args, **kw)
                calls.append(k)
                return k
            os.copy_file_range = spy
        try:
            for part in (b, c):
                with open(part, "rb") as fh:
                    dist._append_file(out, fh)
        finally:
            if real is not None:
                os.copy_file_range = real
    assert a.read_bytes() == b"header\n" + b"x" * 100_000 + b"tail"
    if real is not None:
        assert sum(calls) == 100_004        # every byte went through the in-kernel copy


def test_streamed_bam_equals_resident(tmp_path, monkeypatch):
    """STREAMED ingest (npore_bam_open_mode 2: block table + 22 bytes per record, the records of a batch inflated from
    the BGZF blocks they lie in) against the resident mode and the Python reader: a BAM of ~120 BGZF blocks indexed in
    windows of 2 blocks (records straddle blocks and windows, the header straddles the first window), selection by
    region, packed bases / CIGARs, SAM text, all through tiny scattered batches; the saved record index reopens the
    file without a second pass; a file that is not BGZF is refused in that mode."""
    from npore_amd import bam
    rng = np.random.default_rng(12)
    contigs = {f"c{k}_{'x' * 40}": "".join(rng.choice(list("ACGT"), 60_000)) for k in range(3)}
    fa = tmp_path / "r.fa"
    fa.write_text("".join(f">{n}\n" + "\n".join(s_[i:i + 70] for i in range(0, len(s_), 70)) + "\n" for n, s_ in contigs.items()))
    names = list(contigs)
    recs = []
    for k in range(900):
        rid = int(rng.integers(0, 3))
        L = int(rng.choice([30, 200, 3000, 9000]))
        cig = [(0, L // 2), (1, 2), (2, 3), (7, L - L // 2 - 2)]
        if k % 3 == 0: cig = [(4, 5)] + cig
        if k % 7 == 0: cig = cig + [(4, 4), (5, 6)]
        qlen = sum(n for op, n in cig if op in (0, 1, 4, 7, 8))
        rlen = sum(n for op, n in cig if op in (0, 2, 3, 7, 8))
        pos = int(rng.integers(0, 60_000 - rlen))
        recs.append(dict(name=f"read{k}", flag=[0, 16, 0x100, 0, 4, 0x800][k % 6], ref_id=rid, pos=pos, mapq=int(rng.integers(0, 61)),
                         cigar=cig, seq="".join(rng.choice(list("ACGT"), qlen)),
                         qual=None if k % 5 == 0 else rng.integers(0, 42, qlen).astype(np.uint8).tobytes(), hp=None if k % 4 == 1 else k % 3))
    recs.sort(key=lambda r_: (r_["ref_id"], r_["pos"]))
    bp = str(tmp_path / "big.bam")
    # (+ 2 500 read-less contigs in the header: ~300 KB, so the header itself spans several windows)
    bam.write_bam(bp, [(n, len(s_)) for n, s_ in contigs.items()] + [(f"unplaced_scaffold_{k:05d}_{'y' * 30}", 1000 + k) for k in range(2500)],
                  recs, level=1)
    monkeypatch.setenv("NPORE_BAM_WINDOW_BLOCKS", "2")
    regions = [(names[0], 0, 59_999), (names[2], 10_000, 30_000), (names[1], 0, 59_999)]
    n1 = _native_vs_python(bp, str(fa), regions, stream=True)
    assert n1 == _native_vs_python(bp, str(fa), regions, stream=False) and n1 > 300
    assert _native_vs_python(bp, str(fa), regions, max_reads=17, stream=True) == 17
    a, b = bam.NativeBam(bp, stream=False), bam.NativeBam(bp, stream=True)
    lib = _lib.load()
    assert lib.npore_bam_inflated_size(a.handle) == lib.npore_bam_inflated_size(b.handle)
    ia = a.select(regions)
    assert np.array_equal(ia, b.select(regions))
    nf = bam.NativeFasta(str(fa))
    for sub in (ia[::37], ia[5:9], ia[-3:][::-1], ia[:0]):         # scattered, adjacent, reversed, empty batches
        pa, pb = a.pack(nf, sub), b.pack(nf, sub)
        assert all(np.array_equal(x, y) for x, y in zip(pa, pb))
        fin, st = ["5M"] * len(sub), np.zeros(len(sub), np.int32)
        assert a.format_sam(sub, fin, st) == b.format_sam(sub, fin, st)
    # the record index of a streamed handle, saved and loaded by another handle (what the other ranks of a node do)
    ix = str(tmp_path / "big.idx")
    assert lib.npore_bam_save_index(b.handle, os.fsencode(ix)) == 0 and os.path.getsize(ix) < 400_000 + 22 * b.n_records
    h = lib.npore_bam_open_mode(os.fsencode(bp), 2, 2, os.fsencode(ix))
    assert h and lib.npore_bam_n_records(h) == b.n_records and lib.npore_bam_is_streamed(h) == 1
    lib.npore_bam_close(h)
    assert lib.npore_bam_dump_inflated(b.handle, os.fsencode(str(tmp_path / "no.raw"))) != 0      # nothing to dump
    raw = str(tmp_path / "copy.raw")
    assert lib.npore_bam_dump_inflated(a.handle, os.fsencode(raw)) == 0
    assert not lib.npore_bam_open_mode(os.fsencode(raw), 1, 2, None) and "BGZF" in _lib.last_error()
    # several local ranks: rank 0 indexes and shares the index, rank 1 loads it
    if os.path.isdir("/dev/shm"):
        monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
        monkeypatch.setenv("MASTER_PORT", str(41000 + os.getpid() % 20000))
        monkeypatch.setenv("LOCAL_RANK", "0")
        maker = bam.NativeBam(bp, stream=True)
        assert maker._shared and maker._shared[0].endswith(".idx") and os.path.exists(maker._shared[0])
        monkeypatch.setenv("LOCAL_RANK", "1")
        user = bam.NativeBam(bp, stream=True)
        assert user.streamed and np.array_equal(user.select(regions), ia)
        shared = maker._shared[0]
        user.close(); maker.close()
        assert not os.path.exists(shared)
        os.remove(shared.replace(".idx", ".skip"))
    a.close(); b.close(); nf.close()


def _asm_checker():
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_asm_pending", os.path.join(REPO, "scripts", "check_asm_pending.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_fill_step_asm_is_generated_and_counter_clean(tmp_path):
    """The committed fill_step_asm.inc is what gen_fill_asm.py generates, and the generated text obeys the architected
    ordering rules on EVERY path (scripts/check_asm_pending.py): nothing touches a register a load is still going to
    fill, nothing writes a data register of a > 64-bit LDS write / store before a wait on its counter has retired it
    (the ds_write_b128 hazard of round 3), the text starts with a full wait and ends with nothing on the LGKM counter."""
    chk = _asm_checker()
    G = chk.G
    out = tmp_path / "gen.inc"
    subprocess.check_call([sys.executable, os.path.join(REPO, "npore_amd", "csrc", "gen_fill_asm.py"), "--out", str(out)],
                          stdout=subprocess.DEVNULL)
    assert out.read_text() == open(os.path.join(REPO, "npore_amd", "csrc", "fill_step_asm.inc")).read()
    for role in range(4):
        assert chk.findings(role) == [], (role, chk.findings(role)[:5])
    # the checker sees the hazards it is there for.  (1) round 3's: LENST (v109) restored right behind the record's
    # ds_write_b128 -- eight idle cycles in between are NOT a guarantee
    for role in range(4):
        lines = G.gen_role(role)
        k = max(i for i, ln in enumerate(lines) if ln.startswith("ds_write_b128"))
        bad = lines[:k + 1] + ["s_nop 7", "v_mov_b32 v109, 0x7f800000"] + lines[k + 1:]
        f = chk.findings(role, bad)
        assert any(rule == "R2" and "v109" in regs for rule, _, _, regs in f), role
    # (2) a VALU write into a register an LDS read is still going to fill
    lines = G.gen_role(2)
    k = next(i for i, ln in enumerate(lines) if ln.startswith("ds_read_b128"))
    dst = re.match(r"ds_read_b128 v\[(\d+):", lines[k]).group(1)
    f = chk.findings(2, lines[:k + 1] + [f"v_mov_b32 v{dst}, 0"] + lines[k + 1:])
    assert any(rule == "R1" for rule, _, _, _ in f)
    # (3) a loop-carried register written while exec is narrowed to lane 0 (the first form of this round's LEN variant)
    lines = G.gen_role(2)
    k = next(i for i, ln in enumerate(lines) if ln == "s_mov_b64 exec, %[ml0]")
    f = chk.findings(2, lines[:k + 1] + ["v_mov_b32 v109, 0x7f800000"] + lines[k + 1:])
    assert any(rule == "R4" for rule, _, _, _ in f)
    # (4) no wait at the text's start / its end
    assert any(rule == "R3" for rule, _, _, _ in chk.findings(0, G.gen_role(0)[1:]))
    assert any(rule == "R3" for rule, _, _, _ in chk.findings(1, [ln for ln in G.gen_role(1)][:-1]))


def test_bam_shares_begin_at_record_starts(tmp_path):
    """npore_bam_set_share (several ranks, each one pass over its stretch of the file): the cut points come from the .bai
    LINEAR index and are record starts; the stretches of the ranks tile the record stream; a .bai of another file, no
    .bai, a handle that is not on a BGZF file: refused.  (Host logic only; the walk itself is a GPU test.)"""
    import struct
    recs, pos = [], 0
    rng = np.random.default_rng(3)
    for k in range(400):
        n = int(rng.integers(200, 900))
        seq = "".join(rng.choice(list("ACGT"), n))
        recs.append(dict(name=f"r{k}", flag=0, ref_id=k // 300, pos=pos % 200_000, cigar=[(0, n)], seq=seq, qual=bytes(rng.integers(0, 94, n, dtype=np.uint8)), hp=None))
        pos += int(rng.integers(100, 1200))
        if k == 299:
            pos = 0
    path = str(tmp_path / "s.bam")
    bam.write_bam(path, [("a", 300_000), ("b", 300_000)], recs, level=1)
    h = bam.NativeBam(path, one_pass=True, share=False)
    with pytest.raises(bam.OnePassUnsupported):
        h.set_share(0, 2)
    bai = bam.write_bai(path)
    data = bam._bgzf_decompress(path)
    l_text, = struct.unpack_from("<i", data, 4)
    q = 8 + l_text
    n_ref, = struct.unpack_from("<i", data, q); q += 4
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", data, q); q += 8 + l_name
    starts = []
    while q + 4 <= len(data):
        bs, = struct.unpack_from("<i", data, q)
        starts.append(q); q += 4 + bs
    assert len(starts) == 400
    starts = np.array(starts)
    for world in (1, 2, 3, 7, 16):
        prev_end, total = None, 0
        for rank in range(world):
            has, b0, e0, blk = h.set_share(rank, world)
            assert has == (world > 1)
            if b0 == -1:                                         # nothing left for this rank (and for the ones behind it)
                assert e0 == -1 and prev_end == -1
                continue
            b0 = int(starts[0]) if b0 == 0 else b0
            assert b0 in starts and (e0 == -1 or e0 in starts)
            assert prev_end is None or prev_end == b0
            prev_end = e0
            total += int(((starts >= b0) & (starts < (e0 if e0 != -1 else 1 << 62))).sum())
        assert prev_end == -1 and total == 400, world
    # a one-pass handle learns which contigs have reads from the .bai (both here); without one it assumes all do
    h2 = bam.NativeBam(path, one_pass=True, share=False)
    assert h2.refs_with_reads() == {0, 1}
    h2.close()
    recs_one = [dict(r, ref_id=0) for r in recs[:50]]
    p1 = str(tmp_path / "one.bam")
    bam.write_bam(p1, [("a", 300_000), ("b", 300_000), ("c", 10)], recs_one, level=1)
    hx = bam.NativeBam(p1, one_pass=True, share=False)
    assert hx.refs_with_reads() == {0, 1, 2}                     # no index: unknown
    hx.close()
    bam.write_bai(p1)
    hx = bam.NativeBam(p1, one_pass=True, share=False)
    assert hx.refs_with_reads() == {0}
    hx.close()
    # the golden BAM's own .bai (made by samtools: bins AND a linear index)
    g = bam.NativeBam(os.path.join(GOLDEN, "data", "reads.bam"), one_pass=True, share=False)
    assert g.set_share(0, 2)[0] == 1 and g.set_share(1, 2)[0] == 1
    with pytest.raises(bam.OnePassUnsupported):
        g.set_share(1, 2, bai=bai)                               # another file's index: its offsets are not block starts here
    g.close()
    h.close()
