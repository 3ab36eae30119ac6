#!/bin/bash
# GPU box helper: the default bench line (no extra legs) under a list of environment settings, one line each.
# usage: bash tests/sweep_env.sh <tag> "VAR=a VAR=b 'VAR=c OTHER=d'" ...   -> gpurun_out/<tag>/sweep.log
TAG=$1; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
for setting in "$@"; do
  env $setting python $R/bench.py --steps 2 --warmup 1 --no-extras > $OUT/sweep_last.json 2>$OUT/sweep_last.err
  python - "$setting" $OUT/sweep_last.json <<'PY' | tee -a $OUT/sweep.log
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
p = d["pfp"]["last_step_ms"]
print("%-34s step %7.1f ms  run %7.1f  sort stage %7.1f  dict_sa %6.1f parse_sa %6.1f groups %6.1f distinct %6.1f lists %5.1f" % (
    sys.argv[1], d["ms_per_step"], 1e3 * d["phase_s_avg"]["run"], d["stage_ms_avg"]["suffix_sort"], p["dictionary_sa"], p["parse_sa"],
    p["dictionary_groups"], p["distinct_phrases"], p["lists_emit"]))
PY
done
