#!/bin/bash
# GPU box helper: rocprofv3 kernel statistics of ONE whole-genome share run (tests/big_share_real.py or tests/big_share.py or
# tests/big_c5_all.py), summarised by tests/kstats.py.   usage: bash tests/profile_round6_share.sh <tag> <script and arguments ...>
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
timeout 1500 rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG --output-format csv -- python "$@" > $OUT/run.log 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python $R/tests/kstats.py $OUT/kernel_stats.csv 1 45 > $OUT/kernel_summary.txt
rm -rf $OUT/trace
head -30 $OUT/kernel_summary.txt
grep "^{" $OUT/run.log | tail -3 | cut -c1-700
