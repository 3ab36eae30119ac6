# C3 stand-in through both producers, with memory and stage times (GPU box helper)
cd $GRAFT_REPO_ROOT
python bench.py --steps 2 --warmup 1 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('pfp   ', d['ms_per_step'], d['stage_ms_avg'], d['device_memory'], d['config'].get('stream_producer'), d['config'].get('output_rows'))"
MUMEMTO_PRODUCER=guided MMT_GUIDED_STATS=1 python bench.py --steps 1 --warmup 1 --no-extras 2>gpurun_out/c3_guided.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('guided', d['ms_per_step'], d['stage_ms_avg'], d['device_memory'], d['config'].get('stream_producer'), d['config'].get('output_rows'))"
grep guided gpurun_out/c3_guided.err | tail -5
