#!/bin/bash
# The bench lines committed under profiles/ for a round: the default line and the other configurations.
# usage (GPU box): scripts/final_lines.sh <tag>
tag=${1:-r05}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
o=gpurun_out/$tag; mkdir -p $o
python bench.py --steps 20 --warmup 5 > $o/bench.log 2>&1 || exit 1
grep -h '^{"metric' $o/bench.log | tail -n 1 > $o/bench_line.json
X="--pcie-steps 0 --sustain 0 --production 0 --cpu-sample 16 --cpu-threads 16"     # (with the CPU legs: every CPU string compared with the GPU's)
: > $o/config_lines.jsonl
python bench.py --reads 1000 --band 30 --steps 20 --warmup 5 $X > $o/c2r30.log 2>&1 && grep -h '^{"metric' $o/c2r30.log | tail -n 1 >> $o/config_lines.jsonl
python bench.py --reads 4000 --band 30 --steps 20 --warmup 5 $X > $o/r30x4000.log 2>&1 && grep -h '^{"metric' $o/r30x4000.log | tail -n 1 >> $o/config_lines.jsonl
python bench.py --reads 8000 --band 30 --steps 20 --warmup 5 $X > $o/r30x8000.log 2>&1 && grep -h '^{"metric' $o/r30x8000.log | tail -n 1 >> $o/config_lines.jsonl
python bench.py --reads 100000 --mixed --base-seed 3 --steps 3 --warmup 1 $X > $o/c3.log 2>&1 && grep -h '^{"metric' $o/c3.log | tail -n 1 >> $o/config_lines.jsonl
python bench.py --reads 256 --ref-len 50000 --band 200 --base-seed 5 --steps 5 --warmup 2 $X --cpu-sample 4 > $o/c5.log 2>&1 && grep -h '^{"metric' $o/c5.log | tail -n 1 >> $o/config_lines.jsonl
wc -l $o/config_lines.jsonl
# file to file (BAM -> realigned SAM): 96 000 reads one-pass + indexed (resident, streamed), then a larger file one-pass only
: > $o/realign_lines.jsonl
# (1) the ONT-like file: every read distinct, one uniform quality per base (the reference's fixture generator); (2) rounds 3 - 4's
# file for continuity: 4 000 distinct reads x 24, constant qualities; (3) a larger ONT-like file, one pass, text discarded
python scripts/bench_realign.py --reads 96000 --batch 4000 --py-reads 0 > $o/realign_96k.log 2>&1 && grep -h '^{"metric' $o/realign_96k.log | tail -n 1 >> $o/realign_lines.jsonl
python scripts/bench_realign.py --reads 96000 --batch 4000 --py-reads 0 --distinct 4000 --const-qual > $o/realign_96k_r04file.log 2>&1 && grep -h '^{"metric' $o/realign_96k_r04file.log | tail -n 1 >> $o/realign_lines.jsonl
python scripts/bench_realign.py --reads 300000 --batch 4000 --one-pass-only > $o/realign_300k.log 2>&1 && grep -h '^{"metric' $o/realign_300k.log | tail -n 1 >> $o/realign_lines.jsonl
wc -l $o/realign_lines.jsonl
