#!/bin/bash
# What a user of the command line sees: `python -m npore_amd.realign` on the generated ONT-like BAM, wall time of the whole
# process (interpreter start, context, one pass, SAM written), twice (the second run finds the files in the page cache and
# the driver's memory scrubbed or not as it pleases).
# usage (GPU box): scripts/bench_cli.sh [reads=96000] [dir=/dev/shm/npore_cli]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
n=${1:-96000}; d=${2:-/dev/shm/npore_cli}
mkdir -p $d gpurun_out
python scripts/bench_realign.py --gen-into $d --reads $n > $d/gen.json || exit 1
for rep in 1 2; do
    rm -f $d/out.sam
    s=$(date +%s.%N)
    python -m npore_amd.realign --bam $d/reads.bam --ref $d/ref.fa --out_prefix $d/out > $d/cli_$rep.log 2>&1 || { tail -5 $d/cli_$rep.log; exit 1; }
    e=$(date +%s.%N)
    python3 -c "import sys,os; n=int(sys.argv[1]); t=float(sys.argv[3])-float(sys.argv[2]); print('realign CLI run %s: %d reads in %.2f s = %.0f reads/s, SAM %.2f GB' % (sys.argv[4], n, t, n/t, os.path.getsize(sys.argv[5])/1e9))" $n $s $e $rep $d/out.sam | tee -a gpurun_out/bench_cli.txt
done
rm -rf $d
