#!/bin/bash
# GPU box: both traceback kernels over a few batch shapes (one line per shape and kernel).  usage: scripts/tb_crossover.sh "r:reads r:reads ..."
set -u
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/tbx; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in $1; do
  r=${c%%:*}; n=${c##*:}
  for mode in 1 2; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python3 $R/scripts/exp_tb.py $r $n $mode > $out/m.log 2>&1
    f=$(find $out/t -name "*kernel_stats.csv" | head -1)
    echo "r=$r reads=$n mode=$mode $(grep -i "traceback" $f | cut -d, -f1,4)" | tee -a $out/summary.txt
    rm -rf $out/t
  done
done
