"""Stage times of one C2 batch through a given build of the library (perf experiments:
    python scripts/exp_pmc.py path/to/libnpore_amd.so   -- also the program to put behind `rocprofv3 --pmc ... --`)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npore_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import ctypes
_probe = ctypes.CDLL(_lib.LIB_PATH)          # older builds lack the newest entry points: bind what is there
for _name in list(_lib.SIGNATURES):
    if not hasattr(_probe, _name):
        del _lib.SIGNATURES[_name]
from npore_amd import aln, synth
sub, nps, _, _ = aln.load_default_tables()
ctx = aln.Context(sub, nps)
_r, _n = (int(x) for x in os.environ.get('PMC_CASE', '100:1000').split(':'))
refs, seqs, cigs = synth.make_batch(2, _n, ref_len=10000)
for rep in range(2):
    out, st = ctx.align_batch(refs, seqs, cigs, r=_r, return_status=True)
t = ctx.timing()
print('fill', round(t['fill_ms'], 2), 'prep', round(t['dev_prep_ms'], 2), 'tb', round(t['traceback_ms'], 2))
