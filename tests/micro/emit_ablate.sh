#!/bin/bash
# GPU box: k_emit with one phase cut out at a time (MMT_EMIT_ABLATE; wrong output, timing only) on the C3 stand-in
mkdir -p gpurun_out
for a in 0 1 2 3 4 5; do
  MMT_EMIT_ABLATE=$a python bench.py --no-extras --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]) if l else None
print('ablate $a', d['stage_ms_avg']['stream_windows_in_suffix_sort'] if d else 'failed')"
done
