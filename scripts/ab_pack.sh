#!/bin/bash
# GPU box: the one-pass file pipeline with align()'s inputs unpacked on the device / packed on the host, alternating on one box.
# usage: scripts/ab_pack.sh [reads=96000] [rounds=3]
reads=${1:-96000}; rounds=${2:-3}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
o=gpurun_out/ab_pack; mkdir -p $o
for k in $(seq 1 $rounds); do
  for dp in 1 0; do
    NPORE_DEVICE_PACK=$dp python scripts/bench_realign.py --reads $reads --batch 4000 --py-reads 0 --one-pass-only > $o/run_${dp}_$k.log 2>&1 || exit 1
    python - $o/run_${dp}_$k.log $dp <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith('{"metric')][-1]
e = json.loads(l); op = e["one_pass"]
print("device_pack", sys.argv[2], "reads/s", round(op["reads_per_s"]), "stage sums", op["stage_sums_s"], "rss", op.get("peak_rss_mb"))
PY
  done
done
