"""Process-global configuration and encodings.

Mirrors the role of reference src/cfg.py:1-34: `args` is the argparse Namespace
that realign.py stores here and that align()/get_np_info() read `max_n` /
`max_l` from (reference src/aln.pyx:207-208, 436-437).  Base and CIGAR-op codes
are the reference's (src/cfg.py:11-32) because they are part of the data
format crossing the C-ABI (include/npore_amd.h).
"""
import argparse
from collections import defaultdict

# set by realign.py (or by a caller) -- defaults are the reference CLI defaults
# (src/realign.py:46-51)
args = argparse.Namespace(max_n=6, max_l=100)

bases = "NACGT"
symbols = "NACGT-"
nbases = len(bases)
base_dict = defaultdict(int)
for _i, _c in enumerate("NACGT"):
    base_dict[_c] = _i
    base_dict[_c.lower()] = _i
base_dict["-"] = 5

cigars = "MIDNSHP=XB"
cigar_dict = {c: i for i, c in enumerate(cigars)}

__version__ = "0.1.1+mi355x.1"
