#!/bin/bash
# usage: tools/bench_sweep.sh "<label> <bench args>" ...   -> one summary line per run (timing experiments)
for cfg in "$@"; do
  python bench.py --no-cpu-baseline $cfg > gpurun_out/bench_sweep.log 2>&1
  tail -1 gpurun_out/bench_sweep.log | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('$cfg', '->', d['value'], 'fps', d['ms_per_step'], 'ms', d['config']['stage_ms_per_step'])
except Exception as e:
    print('$cfg', 'FAILED', e)
"
done
