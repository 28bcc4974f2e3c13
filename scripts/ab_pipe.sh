#!/bin/bash
# GPU box: the one-pass file pipeline under several environment settings, alternating on one box.
# usage: scripts/ab_pipe.sh <reads> <rounds> "VAR=a VAR2=b" "VAR=c" ...      (each quoted argument = one setting)
reads=$1; rounds=$2; shift 2
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
o=gpurun_out/ab_pipe; mkdir -p $o
for k in $(seq 1 $rounds); do
  j=0
  for setting in "$@"; do
    j=$((j + 1))
    env $setting python scripts/bench_realign.py --reads $reads --batch 4000 --py-reads 0 --one-pass-only > $o/run_${j}_$k.log 2>&1 || exit 1
    python - $o/run_${j}_$k.log "$setting" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith('{"metric')][-1]
e = json.loads(l); op = e["one_pass"]
print(f"[{sys.argv[2]}]", "reads/s", round(op["reads_per_s"]), "stage sums", op["stage_sums_s"])
PY
  done
done
