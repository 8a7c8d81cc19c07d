#!/bin/bash
# usage: tools/probe_bench_flags.sh <debug-flags> ...   - short default bench runs with engine debug flags (timing experiments)
for f in "$@"; do
  python bench.py --no-cpu-baseline --no-host-fed --no-other-configs --steps 4 --debug-flags $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flags $f', d['value'], 'k_describe', d['config']['stage_ms_per_chunk']['k_describe'])"
done
