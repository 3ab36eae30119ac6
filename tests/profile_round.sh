#!/bin/bash
# GPU box helper: kernel trace + the two PMC passes for k_scan of the default bench (separate rocprofv3 runs).
# usage: bash tests/profile_round.sh <tag>
TAG=${1:-round1_d}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG --output-format csv -- \
    python $R/bench.py --steps 3 --warmup 1 --cpu-sample-bp 0 > $OUT/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex 'mmt::k::k_scan' -d $OUT/pmc_$c -o $c --output-format csv -- \
      python $R/bench.py --steps 2 --warmup 1 --cpu-sample-bp 0 > $OUT/pmc_$c.log 2>&1
done
find $OUT -name "*.csv" | head -20
tail -c 600 $OUT/bench.json
