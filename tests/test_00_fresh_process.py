"""GPU tests (-m gpu) that start their own processes, collected FIRST (file name) so that this pytest
process has not touched the GPU yet when it launches them:

  * bench.py under torch.distributed.run with 2 ranks on this box's one GPU -- the N > 1 path (reads dealt
    by index, per-rank contexts, counter reduction, rank-0 JSON line) executed once under the driver;
  * aln.get_np_info() as the very first call of a fresh process (no align(), no Context, no tables) on a
    5 Mbp sequence, as the reference's callers use it (src/bed.py:62, src/bam.pyx:381).
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_bench_two_ranks_on_one_gpu():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29600 + os.getpid() % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(REPO, "bench.py"),
           "--gpus", "2", "--reads", "96", "--ref-len", "3000", "--band", "30", "--steps", "2", "--warmup", "1",
           "--sustain", "0", "--pcie-steps", "1", "--no-cpu", "--production", "0.05", "--production-reads", "64"]
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                      # rank 0 prints, the other rank stays silent
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["bad_reads"] == 0
    assert line["production_default"]["value"] > 0 and "r=30" in line["production_default"]["workload"]
    assert line["value"] > 0 and line["unit"] == "reads/s" and line["scaling"] == "weak"
    assert line["config"]["reads_per_gpu"] == 96
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"] is None      # CPU baseline is an N=1 item
    assert line["value_pcie_inclusive"]["value"] > 0
    # two ranks on ONE card: the reduction goes over gloo (RCCL needs a device per rank), host-side generation is reported
    # and lies before the timed region
    assert line["config"]["reduction_backend"].startswith("gloo") and line["config"]["host_setup"]["generate_s_max_over_ranks"] > 0
    # (3 kb reads have no committed digests: the main leg compares nothing; the production leg's 2 x 64 reads are r30's 0 ... 127)
    assert line["parity"]["main"]["strings_compared"] == 0
    assert line["parity"]["production_default"] == {"strings_compared": 128, "strings_bad": 0}


def test_bench_four_ranks_every_rank_proves_its_strings():
    """The driver's scaling command at N = 4 with all ranks on this box's one card: `bench.py --gpus 4` at the C2 shape
    (250 reads per rank, r=100) -- EVERY rank compares EVERY string with the committed digest of the pinned oracle (rank
    k holds read indices k, k + 4, ...: 0 ... 999 of `c2`), and the production leg's reads (r=30) likewise.  Four ranks,
    not eight: the pool allows six processes on the card at once, and this pytest process is one of them."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29700 + os.getpid() % 100
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(REPO, "bench.py"),
           "--gpus", "4", "--reads", "250", "--steps", "2", "--warmup", "1", "--sustain", "0", "--pcie-steps", "0",
           "--no-cpu", "--production", "0.05", "--production-reads", "100", "--solo-steps", "1"]
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 4 and line["bad_reads"] == 0
    assert line["parity"]["main"] == {"strings_compared": 1000, "strings_bad": 0}
    assert line["parity"]["production_default"] == {"strings_compared": 400, "strings_bad": 0}
    assert line["parity"]["strings_compared"] == 1400 and line["parity"]["strings_bad"] == 0


def test_bench_two_ranks_production_shape_bookkeeping():
    """`bench.py --gpus 2 --reads 4000 --band 30` with both ranks on this box's one card (the 8-GPU run's command at
    N = 2): value = the reads of BOTH ranks / the slowest rank's time, i.e. 2 x reads_per_gpu x steps / (ms_per_step x
    steps); every rank's strings clean."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29850 + os.getpid() % 100
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(REPO, "bench.py"),
           "--gpus", "2", "--reads", "4000", "--band", "30", "--steps", "3", "--warmup", "1",
           "--sustain", "0", "--pcie-steps", "0", "--no-cpu", "--production", "0", "--solo-steps", "1"]
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["reads_per_gpu"] == 4000 and line["bad_reads"] == 0
    assert abs(line["value"] - 2 * 4000 / (line["ms_per_step"] * 1e-3)) < 0.01 * line["value"]
    assert line["config"]["parallelism"] == "reads x2" and line["scaling"] == "weak"
    # rank k holds reads k, k + 2, ... of seed 2 at r=30: indices 0 ... 7 999, of which `r30` holds the first 4 000 and
    # every 7th after (4 000 + 572)
    assert line["parity"]["main"]["strings_bad"] == 0 and line["parity"]["main"]["strings_compared"] == 4000 + len(range(4000, 8000, 7))


_NP_INFO_FIRST = r"""
import sys, numpy as np
sys.path.insert(0, {repo!r})
from npore_amd import aln, synth
rng = np.random.Generator(np.random.PCG64(77))
seq, _, _ = synth.make_ref(rng, 5_000_000, 0.05)
seq = seq.copy()
seq[1_000_000:1_003_000] = 0                    # an assembly gap
seq[2_000_000:2_000_700] = 3                    # a homopolymer far beyond max_l
seq[3_000_000:3_001_200] = np.tile(np.array([1, 2, 4], np.uint8), 400)
got = aln.get_np_info(seq)                      # first library call of the process: no align(), no tables
assert got.shape == (5_000_000, 2, 6) and got.dtype == np.int32
import oracle
want = np.asarray(oracle.get_np_info(seq))
assert np.array_equal(got, want), np.argwhere(got != want)[:5]
short = aln.get_np_info(np.array([1, 4, 1, 4, 1, 4, 1, 4, 4, 4, 4, 4, 4, 1, 1, 1, 3, 2, 3, 2, 3, 2], np.uint8))
assert short[:, 0, 0].tolist() == [0, 0, 0, 0, 0, 0, 0, 6, 6, 6, 6, 6, 6, 3, 3, 3, 0, 0, 0, 0, 0, 0]
assert aln.get_np_info(np.zeros(0, np.uint8)).shape == (0, 2, 6)
print("NP_INFO_OK", int(got[:, 0, :].max()))
"""


def test_get_np_info_first_call_of_a_fresh_process():
    out = subprocess.run([sys.executable, "-c", _NP_INFO_FIRST.format(repo=REPO)], cwd=REPO,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "NP_INFO_OK 100" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


_ASYNC = r"""
import sys, numpy as np
sys.path.insert(0, {repo!r})
import torch                                   # first: the library then binds to the HIP runtime torch has loaded
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
from bench import pack
from npore_amd import _lib, aln, synth
sub, nps, _, _ = aln.load_default_tables()
ctx = aln.Context(sub, nps, max_n=6, max_l=100, device=0)
lib = _lib.load()
batches, want = [], []
for k in range(6):
    refs, seqs, cigs = synth.make_batch(60 + k, 40 + 7 * k, ref_len=2500 + 300 * k)
    want.append(ctx.align_batch(refs, seqs, cigs, r=30 if k % 2 else 100))        # synchronous host-buffer path
    rb, ro = pack(refs); sb, so = pack(seqs); cb, co = pack(cigs)
    oo = np.zeros(len(refs) + 1, np.int64)
    np.cumsum([len(a) + len(b) for a, b in zip(refs, seqs)], out=oo[1:])
    t = [torch.from_numpy(x).to(dev) for x in (rb, ro, sb, so, cb, co, oo)]
    out = torch.zeros(int(oo[-1]) + 64, dtype=torch.uint8, device=dev)
    ln = torch.zeros(len(refs), dtype=torch.int64, device=dev)
    st = torch.zeros(len(refs), dtype=torch.int32, device=dev)
    batches.append((len(refs), t, out, ln, st, oo))
torch.cuda.synchronize()
before = ctx.total_timing()["launches"]
for k, (n, t, out, ln, st, oo) in enumerate(batches):       # six batches enqueued back to back, two in flight
    rc = lib.npore_align_batch_device(ctx.handle, n, *[x.data_ptr() for x in t[:6]], 5.0, 1.0, 20000, 30 if k % 2 else 100,
                                      out.data_ptr(), t[6].data_ptr(), ln.data_ptr(), st.data_ptr(), None, 0)
    assert rc == 0, _lib.last_error()
ctx.wait()
assert ctx.total_timing()["launches"] - before == 6
for (n, t, out, ln, st, oo), w in zip(batches, want):
    o, l = out.cpu().numpy(), ln.cpu().numpy()
    assert not st.cpu().numpy().any()
    assert [o[oo[i]:oo[i] + l[i]].tobytes().decode() for i in range(n)] == w
# a stream of the caller's: the batch is ordered behind the work it holds, later work on it sees the results
n, t, out, ln, st, oo = batches[0]
s2 = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s2):
    out.zero_(); ln.zero_()
    rc = lib.npore_align_batch_device(ctx.handle, n, *[x.data_ptr() for x in t[:6]], 5.0, 1.0, 20000, 100,
                                      out.data_ptr(), t[6].data_ptr(), ln.data_ptr(), st.data_ptr(), s2.cuda_stream, 0)
    assert rc == 0, _lib.last_error()
    total = ln.sum()                 # enqueued on s2 behind the batch
s2.synchronize()
assert int(total.item()) == sum(len(x) for x in want[0])
ctx.wait()
# one call, several groups: a small traceback budget splits 60 reads into groups that go through the same pipeline
ctx.set("tb_budget_mb", 8)
refs, seqs, cigs = synth.make_batch(70, 60, ref_len=3000)
got = ctx.align_batch(refs, seqs, cigs, r=100)
assert ctx.timing()["launches"] > 3
ctx.set("tb_budget_mb", 0)
assert got == ctx.align_batch(refs, seqs, cigs, r=100)
# host buffers, asynchronously (npore_align_batch_async): page-locked arrays, three batches enqueued back to back, each
# group uploading / downloading its own slice; results complete after the wait
hb = []
for k in (1, 3, 5):
    refs, seqs, cigs = synth.make_batch(60 + k, 40 + 7 * k, ref_len=2500 + 300 * k)
    rb, ro = pack(refs); sb, so = pack(seqs); cb, co = pack(cigs)
    oo = np.zeros(len(refs) + 1, np.int64)
    np.cumsum([len(a) + len(b) for a, b in zip(refs, seqs)], out=oo[1:])
    pin = [torch.from_numpy(x).pin_memory() for x in (rb, sb, cb)]
    out = torch.zeros(int(oo[-1]) + 64, dtype=torch.uint8).pin_memory()
    ln = torch.zeros(len(refs), dtype=torch.int64).pin_memory()
    st = torch.zeros(len(refs), dtype=torch.int32).pin_memory()
    hb.append((k, len(refs), pin, (ro, so, co, oo), out, ln, st))
ctx.set("tb_budget_mb", 16)          # several groups per batch
for k, n, pin, offs, out, ln, st in hb:
    rc = lib.npore_align_batch_async(ctx.handle, n, pin[0].data_ptr(), offs[0].ctypes.data, pin[1].data_ptr(), offs[1].ctypes.data,
                                     pin[2].data_ptr(), offs[2].ctypes.data, 5.0, 1.0, 20000, 30 if k % 2 else 100,
                                     out.data_ptr(), offs[3].ctypes.data, ln.data_ptr(), st.data_ptr())
    assert rc == 0, _lib.last_error()
ctx.wait()
ctx.set("tb_budget_mb", 0)
for k, n, pin, offs, out, ln, st in hb:
    o, l, oo = out.numpy(), ln.numpy(), offs[3]
    assert not st.numpy().any()
    assert [o[oo[i]:oo[i] + l[i]].tobytes().decode() for i in range(n)] == want[k], k
t = ctx.timing()
assert t["h2d_ms"] > 0 and t["d2h_ms"] > 0
ctx.close()
print("ASYNC_OK")
"""


def test_async_batches_in_flight():
    """npore_align_batch_device with sync=0 (include/npore_amd.h): batches enqueued back to back on one context
    (two in flight, the third call recycles the first one's work buffers), results complete after npore_ctx_wait and
    equal to the synchronous host-buffer path's; a caller's stream is honoured; a multi-group call uses the same
    pipeline.  In its own process because torch (device buffers) has to initialise the HIP runtime first."""
    out = subprocess.run([sys.executable, "-c", _ASYNC.format(repo=REPO)], cwd=REPO, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ASYNC_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
