#!/bin/bash
# Several fuzz legs side by side on the GPU box (the oracle is the slow side and single-threaded): seeds from $1,
# $2 seconds each, every second leg FOCUSED; logs under gpurun_out/.  usage: scripts/fuzz_parallel.sh <seed0> <seconds> [legs=4]
seed0=${1:-500}; secs=${2:-300}; legs=${3:-4}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
pids=()
( while true; do sleep 60; date >> gpurun_out/fuzz_heartbeat.log; done ) &      # the pool kills a command that writes nothing for 7 minutes
hb=$!
for k in $(seq 0 $((legs - 1))); do
  s=$((seed0 + k))
  if (( k % 2 )); then f=focus; else f=""; fi
  python tests/tools/fuzz_gpu.py $secs $s $f > gpurun_out/fuzz_$s.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
kill $hb 2>/dev/null
tail -n 1 gpurun_out/fuzz_$seed0.log
for k in $(seq 1 $((legs - 1))); do tail -n 1 gpurun_out/fuzz_$((seed0 + k)).log; done
exit $rc
