#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06_run10
mkdir -p $out
cd $GRAFT_REPO_ROOT
T=tests/cpp/test_threads
echo "--- every call pooled" | tee -a $out/threads.jsonl
for n in 8 16 32 48; do timeout 120 $T --time $n 2 --pool-threshold 1 | tee -a $out/threads.jsonl; done
echo "--- every call pooled, same image" | tee -a $out/threads.jsonl
for n in 16 32 48; do timeout 120 $T --time $n 2 --pool-threshold 1 --same-image | tee -a $out/threads.jsonl; done
echo "--- default" | tee -a $out/threads.jsonl
for n in 16 32; do timeout 120 $T --time $n 2 | tee -a $out/threads.jsonl; done
timeout 600 python -m pytest tests/test_gpu_round6.py -q -m gpu 2>&1 | tail -3 | tee -a $out/summary.txt
python bench.py --no-other-configs --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); p=d['config']['pcie_fed']; print(d['value'], p['fps'], p['host_to_host_fps'], p['host_to_host_vs_h2d_only'])" | tee -a $out/summary.txt
