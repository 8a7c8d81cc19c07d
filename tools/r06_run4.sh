#!/bin/bash
# round 6, fourth GPU pass: the pool (tests, thread table with and without it), the round-6 tests, soak of the multi-context suite
out=$GRAFT_REPO_ROOT/gpurun_out/r06_run4
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_round6.py -x -q -m gpu > $out/pytest_round6.log 2>&1; echo "pytest round6 rc=$?" | tee -a $out/summary.txt
tail -15 $out/pytest_round6.log | tee -a $out/summary.txt
T=tests/cpp/test_threads
for k in 0 4 1; do
  echo "--- pool threshold $k" | tee -a $out/threads.jsonl
  for n in 1 2 4 8 16; do timeout 120 $T --time $n 2 --pool-threshold $k | tee -a $out/threads.jsonl; done
  echo "--- pool threshold $k, same image" | tee -a $out/threads.jsonl
  for n in 4 8 16; do timeout 120 $T --time $n 2 --same-image --pool-threshold $k | tee -a $out/threads.jsonl; done
done
echo "--- pool threshold 1, same image, 32 threads" | tee -a $out/threads.jsonl
timeout 120 $T --time 32 2 --same-image --pool-threshold 1 | tee -a $out/threads.jsonl
