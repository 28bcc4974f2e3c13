import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from npore_amd import _lib
if os.environ.get('NPORE_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['NPORE_LIB'])
from npore_amd import aln, synth
import oracle
oracle.build()
sub, nps, _, _ = aln.load_default_tables()
for r, nw, p_np, ref_len in [(100,0,0.15,2600),(200,0,0.15,900),(100,1,0.3,3000)]:
    ctx = aln.Context(sub, nps)
    refs, seqs, cigs = synth.make_batch(321, 12, ref_len=ref_len, p_np=p_np)
    got, st = ctx.align_batch(refs, seqs, cigs, r=r, return_status=True)
    nbad = 0
    for k in range(len(refs)):
        want = oracle.align(refs[k], seqs[k], cigs[k], sub, nps, r=r)
        if got[k] != want:
            nbad += 1
            if nbad == 1:
                d = next((i for i in range(min(len(want), len(got[k]))) if want[i] != got[k][i]), -1)
                print(f"  first mismatch read {k}: len got {len(got[k])} want {len(want)} first diff at {d}: got {got[k][max(0,d-10):d+30]} want {want[max(0,d-10):d+30]}")
    print(f"r={r} nw={nw} p_np={p_np} ref_len={ref_len}: {nbad}/{len(refs)} bad, status {st.tolist()}", flush=True)
    ctx.close()
