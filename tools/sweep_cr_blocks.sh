#!/bin/bash
# usage (GPU box, from the repo root): bash tools/sweep_cr_blocks.sh : k_classify_refine blocks per frame (BRISK_CR_BLOCKS) through a short bench run each
root=${GRAFT_REPO_ROOT:-$PWD}
for b in 8 16 24 36 64; do
  export BRISK_CR_BLOCKS=$b
  bash $root/tools/ab_kernel.sh "x" ethzasl_brisk_amd/libbrisk_hip.so > /dev/null 2>&1
  cp $(find $root/gpurun_out/ab_libbrisk_hip -name "*kernel_stats.csv" | head -1) $root/gpurun_out/sweep_cr_$b.csv
  grep -o '"value": [0-9.]*' $root/gpurun_out/ab_libbrisk_hip/run.log | head -1 > $root/gpurun_out/sweep_cr_$b.txt
done
