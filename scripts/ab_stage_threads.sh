#!/bin/bash
# The one-pass file run with the host stages' thread counts varied one by one (inflater / pack / post), and what the cgroup says
# about throttling in each: the lease is a quota, and three stage groups of N threads each can be runnable at once.
# usage (GPU box): scripts/ab_stage_threads.sh [reads]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
o=gpurun_out/stage_threads; mkdir -p $o
n=${1:-96000}
: > $o/summary.txt
run() {   # label, env...
    local label=$1; shift
    env "$@" python scripts/bench_realign.py --reads $n --batch 4000 --one-pass-only > $o/$label.log 2>&1 || return 1
    python3 - $o/$label.log "$label" <<'PY' | tee -a $o/summary.txt
import json, sys
d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric')][-1]
o = d["one_pass"]
print(sys.argv[2], "reads/s", d["value"], "realign_s", o["realign_s"], "cpu us/read", o["host_cpu_us_per_read"], "cpus busy", o["host_cpus_busy"], "throttling", o.get("cgroup_throttling"))
PY
}
run default X=1 || exit 1
run inflate8 NPORE_INFLATE_THREADS=8 || exit 1
run inflate12 NPORE_INFLATE_THREADS=12 || exit 1
run pack4_post4 NPORE_PACK_THREADS=4 NPORE_POST_THREADS=4 || exit 1
run inflate8_pack4_post4 NPORE_INFLATE_THREADS=8 NPORE_PACK_THREADS=4 NPORE_POST_THREADS=4 || exit 1
run inflate12_pack6_post6 NPORE_INFLATE_THREADS=12 NPORE_PACK_THREADS=6 NPORE_POST_THREADS=6 || exit 1
run default_again X=1 || exit 1
