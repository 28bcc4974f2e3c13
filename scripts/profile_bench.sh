#!/bin/bash
# Run on the GPU box via gpurun: kernel trace + PMC passes of the default bench (each --pmc set in its own run,
# with --kernel-trace/--stats only in the first, as the pool requires).
# usage: scripts/profile_bench.sh <tag> [bench args...]
set -u
tag=${1:-r02}; shift || true
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
rm -rf $out          # (a box is fresh, but gpurun merges into what the build container already holds)
mkdir -p $out
B="--no-cpu --pcie-steps 0 --sustain 0 --production 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/bench.py --steps 3 --warmup 1 $B "$@" > $out/bench_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $out/pmc1 -- python3 $R/bench.py --steps 1 --warmup 0 $B "$@" > $out/bench_pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc2 -- python3 $R/bench.py --steps 1 --warmup 0 $B "$@" > $out/bench_pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_VMEM --output-format csv -d $out/pmc3 -- python3 $R/bench.py --steps 1 --warmup 0 $B "$@" > $out/bench_pmc3.log 2>&1
find $out -name "*.csv" | head -30
for f in $(find $out/trace -name "*kernel_stats.csv"); do echo "== $f"; cat $f; done
tail -1 $out/bench_trace.log
