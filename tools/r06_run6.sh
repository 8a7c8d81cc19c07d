#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06_run6
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_round6.py -x -q -m gpu > $out/pytest_round6.log 2>&1; echo "pytest round6 rc=$?" | tee -a $out/summary.txt
tail -5 $out/pytest_round6.log | tee -a $out/summary.txt
T=tests/cpp/test_threads
for k in 0 1; do
  echo "--- pool threshold $k" | tee -a $out/threads.jsonl
  for n in 4 8 16; do timeout 120 $T --time $n 2 --pool-threshold $k | tee -a $out/threads.jsonl; done
  echo "--- pool threshold $k, same image" | tee -a $out/threads.jsonl
  for n in 4 8 16 32; do timeout 120 $T --time $n 2 --same-image --pool-threshold $k | tee -a $out/threads.jsonl; done
done
bash tools/r06_soak.sh
