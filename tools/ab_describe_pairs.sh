#!/bin/bash
# A / B on one box: k_describe with the gathers dealt to lane pairs (shipped, round 5) against one sample's ten gathers per lane
# (build variant DS_NO_PAIRS: python -c "from ethzasl_brisk_amd import build; build.build_variant('libbrisk_nopairs', ['DS_NO_PAIRS'])")
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/describe_pairs
for lib in "" ethzasl_brisk_amd/libbrisk_nopairs.so; do
  echo "== ${lib:-shipped}"
  for r in 1 2; do BRISK_HIP_LIB=$lib timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-host-fed --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench workload', d['value'], 'frames/s, k_describe', d['roofline']['avg_launch_ms'], 'ms per 512 frames, frac', d['roofline']['frac'])"; done
  for c in 5 4 dense30 1; do BRISK_HIP_LIB=$lib timeout 300 python bench.py --config $c --config-seconds 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=list(d.values())[0]; print('config $c', v['value'], v['unit'], 'k_describe', v['stage_ms']['k_describe'], 'ms')"; done
done 2>&1 | tee gpurun_out/describe_pairs/ab.txt
