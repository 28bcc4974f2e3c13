#!/bin/bash
# The default bench line and the other configurations (no file-to-file legs).  usage (GPU box): scripts/final_lines_short.sh <tag>
tag=${1:-r03}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
o=gpurun_out/$tag; mkdir -p $o
python bench.py --steps 20 --warmup 5 > $o/bench.log 2>&1 || exit 1
grep -h '^{"metric' $o/bench.log | tail -n 1 > $o/bench_line.json
X="--no-cpu --pcie-steps 0 --sustain 0 --production 0"
: > $o/config_lines.jsonl
python bench.py --reads 1000 --band 30 --steps 20 --warmup 5 $X > $o/c2r30.log 2>&1 && grep -h '^{"metric' $o/c2r30.log | tail -n 1 >> $o/config_lines.jsonl
python bench.py --reads 4000 --band 30 --steps 20 --warmup 5 $X > $o/r30x4000.log 2>&1 && grep -h '^{"metric' $o/r30x4000.log | tail -n 1 >> $o/config_lines.jsonl
python bench.py --reads 8000 --band 30 --steps 20 --warmup 5 $X > $o/r30x8000.log 2>&1 && grep -h '^{"metric' $o/r30x8000.log | tail -n 1 >> $o/config_lines.jsonl
python bench.py --reads 100000 --mixed --base-seed 3 --steps 3 --warmup 1 $X > $o/c3.log 2>&1 && grep -h '^{"metric' $o/c3.log | tail -n 1 >> $o/config_lines.jsonl
python bench.py --reads 256 --ref-len 50000 --band 200 --base-seed 5 --steps 5 --warmup 2 $X > $o/c5.log 2>&1 && grep -h '^{"metric' $o/c5.log | tail -n 1 >> $o/config_lines.jsonl
wc -l $o/config_lines.jsonl
