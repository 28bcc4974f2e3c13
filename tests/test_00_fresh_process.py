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
           "--sustain", "0", "--pcie-steps", "1", "--no-cpu"]
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                      # rank 0 prints, the other rank stays silent
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["bad_reads"] == 0
    assert line["value"] > 0 and line["unit"] == "reads/s" and line["scaling"] == "weak"
    assert line["config"]["reads_per_gpu"] == 96
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"] is None      # CPU baseline is an N=1 item
    assert line["value_pcie_inclusive"]["value"] > 0


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
