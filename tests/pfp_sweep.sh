#!/bin/bash
# PFP (w, p) sweep on the bench workload (GPU box)
for wp in ${WPS:-"10 100" "10 50" "10 30" "8 30" "8 20" "6 20" "6 12" "5 10" "4 8"}; do set -- ${wp/:/ }
  r=$(python bench.py --steps 2 --warmup 1 --cpu-sample-bp 0 --producer pfp --pfp-w $1 --pfp-p $2 ${EXTRA} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1), round(d['stage_ms_avg']['suffix_sort'],1), d['pfp']['counts'], d['pfp']['last_step_ms'])")
  echo "w=$1 p=$2 -> $r"
done
