#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06_run9
mkdir -p $out
cd $GRAFT_REPO_ROOT
T=tests/cpp/test_threads
for c in 2 4 6 8; do
  echo "--- pool contexts $c, every call pooled" | tee -a $out/threads.jsonl
  for n in 16 32 48; do BRISK_POOL_CONTEXTS=$c timeout 120 $T --time $n 2 --pool-threshold 1 | tee -a $out/threads.jsonl; done
done
echo "--- pool contexts 8, same image, GPU_MAX_HW_QUEUES=8" | tee -a $out/threads.jsonl
for n in 16 32 48; do GPU_MAX_HW_QUEUES=8 BRISK_POOL_CONTEXTS=8 timeout 120 $T --time $n 2 --pool-threshold 1 --same-image | tee -a $out/threads.jsonl; done
