#!/bin/bash
# GPU box helper: rebuild-free sweep is not possible (CAP/TILE are compile-time); prints PFP stage times of bench.py
python bench.py --steps 3 --warmup 1 --cpu-sample-bp 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['pfp']['last_step_ms'], d['pfp']['counts']['oversized_groups'], d['config']['output_rows'])"
