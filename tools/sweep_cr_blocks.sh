for b in 8 16 24 36 64; do
  export BRISK_CR_BLOCKS=$b
  bash tools/ab_kernel.sh "x" ethzasl_brisk_amd/libbrisk_hip.so > /dev/null 2>&1
  cp $(find gpurun_out/ab_libbrisk_hip -name "*kernel_stats.csv" | head -1) gpurun_out/sweep_cr_$b.csv
  grep -o '"value": [0-9.]*' gpurun_out/ab_libbrisk_hip/run.log | head -1 > gpurun_out/sweep_cr_$b.txt
done
