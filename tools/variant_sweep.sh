#!/bin/bash
# usage: tools/variant_sweep.sh "<hipcc -D flags>" ... ; rebuilds the library per variant on the GPU box and runs the bench
for v in "$@"; do
  BRISK_HIPCC_EXTRA="$v" python -c "from ethzasl_brisk_amd import build; build.build(force=True)" > gpurun_out/variant_build.log 2>&1 || { echo "$v BUILD FAILED"; tail -5 gpurun_out/variant_build.log; continue; }
  echo "== $v"
  bash tools/bench_sweep.sh "--steps 10"
done
