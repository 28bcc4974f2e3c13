#!/bin/bash
# Kernel trace (no counters) of one bench configuration on the GPU box: per-kernel durations of the whole path.
# usage: scripts/trace_only.sh <tag> [bench args...]   -> gpurun_out/trace_<tag>/kernel_stats.csv
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/trace_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --pcie-steps 0 --sustain 0 --production 0 "$@" > $out/bench.log 2>&1
rc=$?
f=$(find $out/t -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -d, -f1-4 $f | grep -v "at::native\|rocclr" > $out/kernel_stats.csv
rm -rf $out/t
tail -1 $out/bench.log | cut -c1-400
exit $rc
