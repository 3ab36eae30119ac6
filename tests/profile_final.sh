#!/bin/bash
# GPU box helper: default bench line + rocprofv3 kernel trace of the same command + the 94 x 20 Mbp line.
# usage: bash tests/profile_final.sh <tag>      (the PMC passes for k_scan are in tests/profile_round.sh)
TAG=${1:-round1_j}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG --output-format csv -- \
    python $R/bench.py --steps 3 --warmup 1 --cpu-sample-bp 0 > $OUT/trace.log 2>&1
timeout 900 python $R/bench.py --haps 94 --length 20000000 --divergence 0.001 --seed 3 --steps 3 --warmup 1 --cpu-sample-bp 0 \
    > $OUT/bench_94hap_20Mbp.json 2> $OUT/bench_94.err
find $OUT -name "*kernel_stats.csv"
tail -c 300 $OUT/bench.json; tail -c 300 $OUT/bench_94hap_20Mbp.json
