#!/bin/bash
# usage: tools/probe_bench_quick.sh <label>   - one short default bench run, prints value and stage intervals (env knobs / BRISK_HIP_LIB apply)
python bench.py --no-cpu-baseline --no-host-fed --no-other-configs --steps 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['config']['stage_ms_per_chunk'])"
