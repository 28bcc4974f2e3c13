#!/bin/bash
# Run on the GPU box via gpurun: PMC passes (instruction mix + occupancy of the issue slots; LDS pipe) of ONE C2 batch per
# library build given, summarised per fill_kernel launch.   usage: scripts/pmc_ab.sh <outdir> lib1.so lib2.so ...
# (PMC_CASE=r:reads selects another batch: scripts/exp_pmc.py)
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  pass=0
  for ctrs in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
              "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
    pass=$((pass+1))
    rocprofv3 --pmc $ctrs --output-format csv -d $out/$name/p$pass -- python3 $R/scripts/exp_pmc.py $R/$lib > $out/$name.p$pass.log 2>&1
  done
  python3 - $out/$name $name <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "fill_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
launches = 2          # exp_pmc.py repeats the batch: report per launch
print(sys.argv[2], {k: round(v / launches / 1e9, 4) for k, v in sorted(agg.items())}, flush=True)
PY
done
