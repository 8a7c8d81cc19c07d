#!/bin/bash
# round 6, fifth GPU pass: the pool with four engine contexts (tests, thread table), hardware-queue setting, chain priority A/B
out=$GRAFT_REPO_ROOT/gpurun_out/r06_run5
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_round6.py -x -q -m gpu > $out/pytest_round6.log 2>&1; echo "pytest round6 rc=$?" | tee -a $out/summary.txt
tail -15 $out/pytest_round6.log | tee -a $out/summary.txt
T=tests/cpp/test_threads
for k in 4 1; do
  echo "--- pool threshold $k" | tee -a $out/threads.jsonl
  for n in 1 2 4 8 16; do timeout 120 $T --time $n 2 --pool-threshold $k | tee -a $out/threads.jsonl; done
  echo "--- pool threshold $k, same image" | tee -a $out/threads.jsonl
  for n in 4 8 16 32; do timeout 120 $T --time $n 2 --same-image --pool-threshold $k | tee -a $out/threads.jsonl; done
done
echo "--- pool threshold 4, same image, GPU_MAX_HW_QUEUES=8" | tee -a $out/threads.jsonl
for n in 4 8 16 32; do GPU_MAX_HW_QUEUES=8 timeout 120 $T --time $n 2 --same-image --pool-threshold 4 | tee -a $out/threads.jsonl; done
echo "--- pool threshold 4, GPU_MAX_HW_QUEUES=8" | tee -a $out/threads.jsonl
for n in 8 16; do GPU_MAX_HW_QUEUES=8 timeout 120 $T --time $n 2 --pool-threshold 4 | tee -a $out/threads.jsonl; done
bash tools/r06_chainprio.sh
