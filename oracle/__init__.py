"""CPU oracle for the nPoRe align() path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package.  Nothing under npore_amd/ does.
"""
from .oracle import (  # noqa: F401
    build, load, align, align_batch, align_batch_procs, get_np_info, lib_path,
)
