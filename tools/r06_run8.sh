#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06_run8
mkdir -p $out
cd $GRAFT_REPO_ROOT
T=tests/cpp/test_threads
echo "--- default policy" | tee -a $out/threads.jsonl
for n in 1 8 16 24 32 48; do timeout 120 $T --time $n 2 | grep -v "^pool phases" | tee -a $out/threads.jsonl; done
echo "--- never pooled" | tee -a $out/threads.jsonl
for n in 24 32 48; do timeout 120 $T --time $n 2 --pool-threshold 0 | tee -a $out/threads.jsonl; done
echo "--- default, same image" | tee -a $out/threads.jsonl
for n in 16 32; do timeout 120 $T --time $n 2 --same-image | grep -v "^pool phases" | tee -a $out/threads.jsonl; done
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k pool > $out/pytest_pool$i.log 2>&1; echo "pytest pool $i rc=$?" | tee -a $out/summary.txt; tail -3 $out/pytest_pool$i.log | tee -a $out/summary.txt; done
timeout 2400 python -m pytest tests -x -q -m gpu > $out/pytest_all.log 2>&1; echo "pytest all rc=$?" | tee -a $out/summary.txt
tail -5 $out/pytest_all.log | tee -a $out/summary.txt
