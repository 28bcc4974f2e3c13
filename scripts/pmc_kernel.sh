#!/bin/bash
# Run on the GPU box via gpurun: kernel trace + one PMC pass (instruction mix, issue-slot occupancy) of ONE configuration,
# summarised per launch for every kernel whose name contains <pattern>.
#   usage: scripts/pmc_kernel.sh <outdir> <pattern> <r> <reads> [ref_len]      -> gpurun_out/<outdir>/{stats.csv,pmc.txt}
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$1; pat=$2; r=$3; reads=$4; rl=${5:-10000}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python3 $R/scripts/exp_one.py $r $reads 3 $rl > $out/trace.log 2>&1 || exit 1
f=$(find $out/t -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -d, -f1-4 $f | grep -v "at::native\|rocclr" > $out/stats.csv
rm -rf $out/t
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU \
    --output-format csv -d $out/p -- python3 $R/scripts/exp_one.py $r $reads 2 $rl > $out/pmc.log 2>&1 || exit 1
python3 - $out/p "$pat" <<'PY' > $out/pmc.txt
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k, c in agg.items():
    n = len(disp[k])
    print(k, "launches", n, {a: round(v / n / 1e6, 3) for a, v in sorted(c.items())}, "(millions per launch)")
PY
rm -rf $out/p
cat $out/stats.csv | head -20; cat $out/pmc.txt
