#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06_run11
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu -x > $out/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?" | tee -a $out/summary.txt
grep -E "passed|failed" $out/pytest_gpu.log | tail -2 | tee -a $out/summary.txt
T=tests/cpp/test_threads
for n in 1 2 4 8 16; do timeout 100 $T --time $n 4 --pool-threshold 0 | grep "^{" | tee -a $out/threads.jsonl; done
for n in 1 16; do timeout 100 $T --time $n 4 --pool-threshold 0 --same-image | grep "^{" | tee -a $out/threads.jsonl; done
timeout 100 $T --time 1 3 640 480 --pool-threshold 0 | grep "^{" | tee -a $out/threads.jsonl
python3 tools/soak.py threads 300 600 2>&1 | tail -2 | tee -a $out/summary.txt
