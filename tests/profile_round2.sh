#!/bin/bash
# GPU box helper: default bench line (C3 stand-in) + rocprofv3 kernel trace of the same command + the two PMC passes
# (FETCH_SIZE, WRITE_SIZE; counters only, separate runs) for k_scan.
# usage: bash tests/profile_round2.sh <tag> [trace|pmc|all]   -> gpurun_out/<tag>/
TAG=${1:-round2_a}
WHAT=${2:-all}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
if [ "$WHAT" = all ] || [ "$WHAT" = bench ]; then
  timeout 900 python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
  tail -c 400 $OUT/bench.json
fi
if [ "$WHAT" = all ] || [ "$WHAT" = trace ]; then
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG --output-format csv -- \
      python $R/bench.py --steps 2 --warmup 1 --no-extras > $OUT/trace.log 2>&1
  cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
  head -40 $OUT/kernel_stats.csv
fi
if [ "$WHAT" = all ] || [ "$WHAT" = pmc ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex 'mmt::k::k_scan' -d $OUT/pmc_$c -o $c --output-format csv -- \
        python $R/bench.py --steps 1 --warmup 1 --no-extras > $OUT/pmc_$c.log 2>&1
    cp $(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1) $OUT/${c}_counter_collection.csv
  done
  ls -la $OUT/*_counter_collection.csv
fi
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
