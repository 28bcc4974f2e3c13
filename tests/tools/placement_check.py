"""Child process of test_chunkmajor_placement_matches_shipped (tests/test_gpu_parity.py): repeated full launches of C2's
reads with ANOTHER build of the library (NPORE_AMD_LIB), one digest per read of the first launch and, per further
launch, the reads that differ from it.
    NPORE_AMD_LIB=<lib.so> python tests/tools/placement_check.py <out.json> <reps> <r> [<r> ...]"""
import hashlib
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def digests(strings):
    return [hashlib.sha256(s.encode()).hexdigest()[:16] for s in strings]


def main():
    out_path, reps, rs = sys.argv[1], int(sys.argv[2]), [int(x) for x in sys.argv[3:]]
    from npore_amd import _lib, aln, synth
    assert os.environ.get("NPORE_AMD_LIB") and _lib.LIB_PATH == os.environ["NPORE_AMD_LIB"]
    sub, nps, _, _ = aln.load_default_tables()
    ctx = aln.Context(sub, nps, device=0)
    refs, seqs, cigs = synth.make_batch(2, 1000)
    res = {}
    for r in rs:
        first, st = ctx.align_batch(refs, seqs, cigs, r=r, return_status=True)
        assert not st.any()
        differ = []
        for rep in range(reps - 1):
            got = ctx.align_batch(refs, seqs, cigs, r=r)
            differ.append([k for k in range(len(got)) if got[k] != first[k]])
        res[str(r)] = {"first": digests(first), "differ": differ}
        print(f"r={r}: {reps} launches, {sum(1 for d in differ if d)} differ from the first", flush=True)
    ctx.close()
    with open(out_path, "w") as fh:
        json.dump(res, fh)


if __name__ == "__main__":
    main()
