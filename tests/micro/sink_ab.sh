cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export MUMEMTO_NO_TEXT_SINK=1; else unset MUMEMTO_NO_TEXT_SINK; fi
  python bench.py --steps 3 --warmup 1 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('nosink=$v', round(d['ms_per_step'],1), {k:round(x,3) for k,x in d['phase_s_avg'].items()}, {k:round(x,1) for k,x in d['stage_ms_avg'].items()})"
done
