#!/bin/bash
# The 1 / 2 / 4 / 8 GPU table of bench.py on one node (weak scaling: the same batch size per GPU, reads dealt by
# index, one RCCL reduction at the end).  Run on a box with 8 MI355X; with fewer GPUs it stops at what is there
# (more ranks than GPUs still run -- ranks then share devices and the reduction goes over gloo -- but measure nothing).
# Every line is also parity evidence: every rank compares its output strings with the committed digests of the pinned
# oracle (tests/golden/fullsize_digests.npz; bench.py digest_parity) and the counts are asserted here.
#   usage: scripts/run_scale.sh [bench.py arguments, e.g. --steps 20 --warmup 5]
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
ngpu=$(python3 -c "import torch; print(torch.cuda.device_count())")
echo "gpus visible: $ngpu"
printf "%-6s %-14s %-12s %-10s %-16s %-22s %s\n" n_gpus reads/s ms_per_step per_gpu efficiency_vs_1 host_generate_s_max backend
base=
for n in 1 2 4 8; do
  [ "$n" -gt "$ngpu" ] && break
  if [ "$n" -eq 1 ]; then
    line=$(python3 bench.py --gpus 1 --no-cpu --sustain 0 --pcie-steps 0 "$@" | tail -1)
  else
    line=$(python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $((29500 + n)) \
           bench.py --gpus "$n" --no-cpu --sustain 0 --pcie-steps 0 "$@" 2>/dev/null | grep '^{' | tail -1)
  fi
  python3 - "$n" "$line" "${base:-0}" <<'PY'
import json, sys
n, line, base = int(sys.argv[1]), json.loads(sys.argv[2]), float(sys.argv[3])
v = line["value"]
hs = line["config"]["host_setup"]
eff = "" if not base else f"{v / n / base:.3f}"
print(f"{n:<6d} {v:<14.1f} {line['ms_per_step']:<12.2f} {v / n:<10.1f} {eff:<16s} {hs['generate_s_max_over_ranks']:<22.2f} {line['config']['reduction_backend']}")
assert line["n_gpus"] == n and line["bad_reads"] == 0
par = line["parity"]
assert par["strings_bad"] == 0 and par["strings_compared"] > 0, par      # every rank compared its strings with the oracle's digests
print(f"       parity: {par['strings_compared']} strings of {n} rank(s) equal to the pinned oracle's digests")
if n > 1:
    assert line["config"]["reduction_backend"] == "rccl", "ranks shared a device: nothing was measured"
PY
  [ "$n" -eq 1 ] && base=$(python3 -c "import json,sys; print(json.loads(sys.argv[1])['value'])" "$line")
done
