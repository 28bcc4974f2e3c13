#!/bin/bash
# Run on the GPU box via gpurun: kernel trace + PMC passes of the default bench (each --pmc set in its own run,
# with --kernel-trace/--stats only in the first, as the pool requires).
# usage: scripts/profile_bench.sh <tag> [bench args...]
set -u
tag=${1:-r02}; shift || true
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out        # (a box is fresh; locally, clear gpurun_out/prof_<tag> before a re-run: gpurun MERGES new files into it)
B="--no-cpu --pcie-steps 0 --sustain 0 --production 0 --solo-steps 0 --pipeline 0"    # (--pipeline 0: every step complete before the next: each kernel has the GPU to itself, as its duration is quoted)
PMC3=${PMC3:-"WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_VMEM"}   # (a 4 000-read launch: PMC3="WRITE_SIZE", the LDS counters take > 7 min there)
PASSES=${PASSES:-"trace pmc1 pmc2 pmc3 pmc4"}
T="timeout -k 10 ${PASS_TIMEOUT:-300}"      # per pass: a counter pass that crawls (r04: FETCH_SIZE on a 4 000-read launch) must not eat the whole call
( while true; do sleep 60; date >> $out/heartbeat.log; done ) &      # the pool kills a command that writes nothing for 7 minutes
hb=$!
[[ $PASSES == *trace* ]] && $T rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/bench.py --steps 3 --warmup 1 $B "$@" > $out/bench_trace.log 2>&1
[[ $PASSES == *pmc1* ]] && $T rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $out/pmc1 -- python3 $R/bench.py --steps 1 --warmup 0 $B "$@" > $out/bench_pmc1.log 2>&1
[[ $PASSES == *pmc2* ]] && $T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc2 -- python3 $R/bench.py --steps 1 --warmup 0 $B "$@" > $out/bench_pmc2.log 2>&1
[[ $PASSES == *pmc3* ]] && $T rocprofv3 --pmc $PMC3 --output-format csv -d $out/pmc3 -- python3 $R/bench.py --steps 1 --warmup 0 $B "$@" > $out/bench_pmc3.log 2>&1
[[ $PASSES == *pmc4* ]] && $T rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/pmc4 -- python3 $R/bench.py --steps 3 --warmup 1 $B "$@" > $out/bench_pmc4.log 2>&1    # clock held under load = GRBM_GUI_ACTIVE / 8 XCDs / kernel time
kill $hb 2>/dev/null
find $out -name "*.csv" | head -30
for f in $(find $out/trace -name "*kernel_stats.csv"); do echo "== $f"; cat $f; done
tail -1 $out/bench_trace.log
