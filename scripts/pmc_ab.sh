#!/bin/bash
# Run on the GPU box via gpurun: one PMC pass (instruction mix + occupancy of the issue slots) of ONE C2 batch per
# library build given, summarised per fill_kernel launch.   usage: scripts/pmc_ab.sh <outdir> lib1.so lib2.so ...
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU \
      --output-format csv -d $out/$name -- python3 $R/scripts/exp_pmc.py $R/$lib > $out/$name.log 2>&1
  python3 - $out/$name $name <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "fill_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
disp = {k: len({1}) for k in agg}
# two launches per run (exp_pmc.py repeats the batch): report per launch
launches = 2
print(sys.argv[2], {k: round(v / launches / 1e9, 4) for k, v in sorted(agg.items())}, flush=True)
PY
done
