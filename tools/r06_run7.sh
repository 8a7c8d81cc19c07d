#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06_run7
mkdir -p $out
cd $GRAFT_REPO_ROOT
T=tests/cpp/test_threads
echo "--- pool threshold 1" | tee -a $out/threads.jsonl
for n in 4 16; do timeout 120 $T --time $n 2 --pool-threshold 1 | tee -a $out/threads.jsonl; done
echo "--- pool threshold 1, same image" | tee -a $out/threads.jsonl
for n in 4 16 32; do timeout 120 $T --time $n 2 --same-image --pool-threshold 1 | tee -a $out/threads.jsonl; done
timeout 1200 python -m pytest tests/test_gpu_round6.py -x -q -m gpu > $out/pytest_round6.log 2>&1; echo "pytest round6 rc=$?" | tee -a $out/summary.txt
tail -5 $out/pytest_round6.log | tee -a $out/summary.txt
timeout 1500 python3 tools/soak.py threads > $out/soak_threads.log 2>&1; echo "soak threads rc=$?" | tee -a $out/summary.txt; tail -12 $out/soak_threads.log | tee -a $out/summary.txt
