#!/bin/bash
# GPU box: kernel trace of both traceback kernels.  usage: scripts/trace_tb.sh <outdir> <r> <reads>
set -u
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for mode in 1 2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/t$mode -- python3 $R/scripts/exp_tb.py $2 $3 $mode > $out/m$mode.log 2>&1
  f=$(find $out/t$mode -name "*kernel_stats.csv" | head -1)
  echo "mode $mode:"; grep -i "traceback\|gather" $f | cut -d, -f1-4
  rm -rf $out/t$mode
done
