#!/bin/bash
# sweep of the blocks per frame of k_score_blocks (BRISK_SB_BLOCKS) and k_classify_refine (BRISK_CR_BLOCKS): the 64-frame configs and the bench workload
cd "$(dirname "$0")/.." || exit 1
for kn in "128 24" "256 24" "512 24" "256 16" "256 12"; do set -- $kn
  export BRISK_SB_BLOCKS=$1 BRISK_CR_BLOCKS=$2
  for c in dense30 dense50 4; do
    timeout 300 python bench.py --config $c --config-seconds 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=list(d.values())[0]; print('sb $1 cr $2 config $c', v['value'], 'classify stage', v['stage_ms']['k_classify_refine'], 'tie', v['stage_ms']['k_tie_resolve'])"
  done
done
for kn in "0 0" "64 24" "128 24" "32 16" "32 32"; do set -- $kn
  export BRISK_SB_BLOCKS=$1 BRISK_CR_BLOCKS=$2
  for b in 128 512; do timeout 300 python bench.py --no-other-configs --steps 6 --warmup 2 --batch $b 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sb $1 cr $2 bench workload, $b frames per call', d['value'], 'frames/s')"; done
done
