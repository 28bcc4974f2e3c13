import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Build what the tests load (the in-tree gfx950 library, the oracle, the cell model) when a
    compiler is around; on the GPU box the prebuilt files travel with the snapshot."""
    import shutil
    if shutil.which("hipcc"):
        from npore_amd import _lib
        try:
            _lib.build()
        except Exception as e:          # the tests that need it will fail loudly themselves
            print(f"conftest: building libnpore_amd.so failed: {e}", file=sys.stderr)
    if shutil.which("gcc"):
        import oracle
        oracle.build()
        sys.path.insert(0, os.path.join(REPO, "tests"))
        from model import model
        model.build()


@pytest.fixture(scope="session")
def tables():
    """G1: score tables as produced by the reference's calc_score_matrices."""
    z = np.load(os.path.join(GOLDEN, "tables.npz"))
    return z["sub_scores"], z["np_scores"]


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def enc(s):
    d = {"N": 0, "A": 1, "C": 2, "G": 3, "T": 4}
    return np.array([d[c] for c in s.upper()], dtype=np.uint8)


def expand_cigar(c):
    out, n = [], 0
    for ch in c:
        if ch.isdigit():
            n = n * 10 + int(ch)
        else:
            out.append(ch * n)
            n = 0
    return "".join(out)


def collapse_cigar(ex):
    out, last, cnt = [], None, 0
    for ch in ex:
        if ch == last:
            cnt += 1
        else:
            if last is not None:
                out.append(f"{cnt}{last}")
            last, cnt = ch, 1
    if last is not None:
        out.append(f"{cnt}{last}")
    return "".join(out)
