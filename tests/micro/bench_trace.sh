# kernel trace of one bench step with extra bench arguments: bash tests/micro/bench_trace.sh <top> [bench args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TOP=${1:-25}; shift
rocprofv3 --kernel-trace --stats -d /tmp/trb -o t --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-extras "$@" > /tmp/trb.log 2>&1
tail -c 600 /tmp/trb.log | grep -o '"ms_per_step": [0-9.]*\|"stage_ms_avg": {[^}]*}'
python $R/tests/kstats.py $(find /tmp/trb -name "*kernel_stats.csv" | head -1) 1 $TOP
