"""Repeated full launches of one batch with one or more library builds, every launch compared read by read with the
first launch of the first library (tests/tools; used for LABNOTES round 3 "a rare wrong stretch"):
    python tests/tools/race_check.py <r> libA.so [libB.so ...]        (REPS=24 launches per library by default)
Prints, per launch, how many reads differ and where the first difference lies."""
import sys, os, subprocess, json
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
def child(lib, r, reps):
    from npore_amd import _lib
    _lib.LIB_PATH=os.path.abspath(lib)
    from npore_amd import aln, synth
    sub,nps,_,_=aln.load_default_tables()
    ctx=aln.Context(sub,nps)
    refs,seqs,cigs=synth.make_batch(2,1000,ref_len=10000)
    outs=[]
    for k in range(reps):
        out,st=ctx.align_batch(refs,seqs,cigs,r=r,return_status=True)
        outs.append(out)
    json.dump(outs, open(f"/tmp/rc_{os.path.basename(lib)}_{r}.json","w"))
if sys.argv[1]=="child":
    child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4])); sys.exit(0)
r=int(sys.argv[1]); libs=sys.argv[2:]
ref=None
for lib in libs:
    subprocess.run([sys.executable, __file__, "child", lib, str(r), os.environ.get("REPS", "24")], check=True)
    outs=json.load(open(f"/tmp/rc_{os.path.basename(lib)}_{r}.json"))
    if ref is None: ref=outs[0]
    for k,o in enumerate(outs):
        bad=[i for i,(a,b) in enumerate(zip(o,ref)) if a!=b]
        msg=f"{lib} r={r} rep {k}: {len(bad)} reads differ"
        for i in bad[:4]:
            a,b=o[i],ref[i]
            p=next((j for j in range(min(len(a),len(b))) if a[j]!=b[j]), min(len(a),len(b)))
            from collections import Counter
            rest=a[p:]
            msg+=f" | read {i} first diff at {p}/{len(b)} lengot {len(a)} rest {dict(Counter(rest))} tail {a[-24:]!r} wanttail {b[-24:]!r}"
            # where do the strings agree again (common suffix)?
            k=0
            while k < min(len(a),len(b)) and a[-1-k]==b[-1-k]: k+=1
            msg+=f" common-suffix {k}"
        print(msg, flush=True)
