#!/bin/bash
# GPU box helper: the two PMC passes (FETCH_SIZE, WRITE_SIZE; counters only, separate runs) for k_scan on the default
# bench.  usage: bash tests/pmc_scan_round.sh <tag>   -> gpurun_out/<tag>/{FETCH_SIZE,WRITE_SIZE}_counter_collection.csv
TAG=${1:-round1_l}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex 'mmt::k::k_scan' -d $OUT/pmc_$c -o $c --output-format csv -- \
      python $R/bench.py --steps 2 --warmup 1 --cpu-sample-bp 0 > $OUT/pmc_$c.log 2>&1
  cp $(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1) $OUT/${c}_counter_collection.csv
done
ls -la $OUT/*_counter_collection.csv
