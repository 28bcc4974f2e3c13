"""A/B of the fill kernel's wave hand-shake: the default build (workgroup release / acquire fences, kernels.hpp)
against the -DNPORE_RELAXED_SYNC build (in-order LDS service + compiler barriers only: round 1's shortcut), each
run over the SAME randomised reads (tests/tools/fuzz_gpu.py, same seed and time budget) and compared with the
oracle read by read.  Minutes of GPU time: not part of the test suite.
usage: python tests/tools/ab_sync.py [seconds per build = 120] [seed = 31]
(both libraries are built by `python __graft_entry__.py build`; on the GPU box the prebuilt files are used)"""
import os
import re
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from npore_amd import _lib  # noqa: E402

secs = sys.argv[1] if len(sys.argv) > 1 else "120"
seed = sys.argv[2] if len(sys.argv) > 2 else "31"
if shutil.which("hipcc"):
    _lib.build()
    _lib.build(relaxed=True)
res = {}
for name, lib in (("default", os.path.join(REPO, "npore_amd", "libnpore_amd.so")), ("relaxed", _lib.RELAXED_LIB_PATH)):
    if not os.path.exists(lib):
        sys.exit(f"{lib} is missing: python __graft_entry__.py build")
    out = subprocess.run([sys.executable, os.path.join(REPO, "tests", "tools", "fuzz_gpu.py"), secs, seed],
                         env=dict(os.environ, NPORE_AMD_LIB=lib), capture_output=True, text=True)
    tail = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-500:]
    print(f"{name:8s} {tail}", flush=True)
    m = re.search(r"(\d+) rounds, (\d+) reads, (\d+) mismatches", tail)
    res[name] = tuple(int(x) for x in m.groups()) if m else None
ok = all(v is not None and v[2] == 0 for v in res.values())
print("A/B:", "both builds equal the oracle on every read" if ok else "MISMATCH -- see above")
sys.exit(0 if ok else 1)
