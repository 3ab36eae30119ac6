# (w, p) of the parse on the bench collection: bash tests/micro/wp_sweep.sh "14,30 14,24 ..." [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PAIRS=${1:-"14,30 14,24 14,20 12,24 16,30 14,40"}; shift
for wp in $PAIRS; do
  w=${wp%,*}; p=${wp#*,}
  python $R/bench.py --steps 2 --warmup 1 --no-extras --producer pfp --pfp-w $w --pfp-p $p "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('w %s p %s: %.1f ms/step, engine %.1f, pfp %s, peak %.0f GB' % ('$w','$p', d['ms_per_step'], d['stage_ms_avg']['engine_total'], {k: round(v) for k, v in d['pfp']['last_step_ms'].items()}, d['device_memory']['peak']/1e9))"
done
