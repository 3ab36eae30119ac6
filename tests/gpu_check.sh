#!/bin/bash
# GPU box helper: a selection of the -m gpu tests, then (optionally) the kernel trace of the default bench line.
# usage: bash tests/gpu_check.sh <tag> "<pytest args>" [trace]   -> gpurun_out/<tag>/pytest.log (+ kernel_stats.csv)
TAG=${1:-check}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -m pytest $2 -x -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/pytest.log | cut -c1-300
if [ "$3" = trace ]; then
  bash tests/profile_round2.sh $TAG trace > /dev/null 2>&1
  python tests/kstats.py $OUT/kernel_stats.csv 3 ${4:-28}
fi
