#!/bin/bash
# A / B of the tie kernel's forms on ONE box (round 5): BRISK_TR_PAIR = 1 (default: two ties per wave up to 64 frames per
# call), 0 (one tie per wave everywhere), 2 (pairs in all batches); BRISK_TR_PAIR_MODE = 1 (default: adjacent ranks in one
# wave) or 0 (the halves 15 ranks apart); BRISK_TR_PAIR_MIN = ties a layer needs to run in pairs.
# usage: tools/ab_tie_pair.sh            (needs the TR_TIMING / TR_TIMELINE variants for the phase tables, see tools/README.md)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/tie_pair
for p in 1 0 2; do
  export BRISK_TR_PAIR=$p
  echo "== BRISK_TR_PAIR=$p"
  timeout 300 python tools/single_frame_stages.py 2>&1 | grep "device-resident"
  for c in 1 4_uniform_single dense30 dense50 4; do
    timeout 300 python bench.py --config $c --config-seconds 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=list(d.values())[0]; print('config $c', v['value'], v['unit'], 'k_tie_resolve', v['stage_ms']['k_tie_resolve'], 'ms')"
  done
  for b in 64 128 512; do timeout 300 python bench.py --no-other-configs --steps 6 --warmup 2 --batch $b 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench workload, $b frames per call', d['value'], 'frames/s')"; done
  if [ -f ethzasl_brisk_amd/libbrisk_trtiming.so ] && [ $p != 2 ]; then
    for w in 4k vga; do BRISK_HIP_LIB=ethzasl_brisk_amd/libbrisk_trtiming.so timeout 200 python3 tools/tie_phases.py $([ $w = vga ] && echo 70 || echo 80) 1 $w 2>&1 | grep -v amdgpu.ids; done
  fi
  if [ -f ethzasl_brisk_amd/libbrisk_trtl.so ] && [ $p != 2 ]; then
    for w in 4k 1080p vga; do BRISK_HIP_LIB=ethzasl_brisk_amd/libbrisk_trtl.so timeout 200 python3 tools/tie_timeline.py $w 2>&1 | grep -v amdgpu.ids; done
  fi
done 2>&1 | tee gpurun_out/tie_pair/ab.txt
