#!/bin/bash
# the one-pass file run at several host thread counts (the lease is a cgroup quota, not a CPU set: more runnable threads than
# the quota are throttled within each scheduler period)
# usage (GPU box): scripts/ab_threads.sh [reads]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
o=gpurun_out/threads; mkdir -p $o
n=${1:-96000}
echo "nproc $(nproc)  cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  affinity $(python3 -c 'import os;print(len(os.sched_getaffinity(0)))')" | tee $o/summary.txt
grep -c ^processor /proc/cpuinfo | tee -a $o/summary.txt
for t in 8 12 16 24 32 48; do
    python scripts/bench_realign.py --reads $n --batch 4000 --one-pass-only --threads $t > $o/t$t.log 2>&1 || exit 1
    python3 - $o/t$t.log $t <<'PY' | tee -a $o/summary.txt
import json, sys
d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric')][-1]
o = d["one_pass"]
print("threads", sys.argv[2], "reads/s", d["value"], "realign_s", o["realign_s"], "cpu us/read", o["host_cpu_us_per_read"], "cpus busy", o["host_cpus_busy"])
PY
done
