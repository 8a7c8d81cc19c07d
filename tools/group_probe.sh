#!/bin/bash
# GPU box: descriptor half of a 256-frame chunk (debug bit 27) as one launch set, and in groups of 8 g frames with the group's
# integral image written right before its descriptors (fresh in the Infinity Cache) or all integral images first (stale).
cd "$(dirname "$0")/.."
run() { python bench.py --no-cpu-baseline --no-host-fed --no-other-configs --steps 6 --warmup 2 --debug-flags "$1" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-12s %-28s %8.3f ms per 256 frames   %s' % ('$1', '$2', 256e3 / d['value'], {k: round(v, 3) for k, v in d['config'].get('stage_ms_per_chunk', {}).items() if 'integral' in k or 'describe' in k}))"; }
run 0x08000000 "descriptor half, one group"
for g in 1 2 4 8; do
  run $(printf '0x%x' $((0x08000000 | g << 20))) "groups of $((8 * g)), fresh"
  run $(printf '0x%x' $((0x08080000 | g << 20))) "groups of $((8 * g)), stale"
done
