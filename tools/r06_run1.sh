#!/bin/bash
# round 6, first GPU pass: the host-exit tests, the thread table of the drop-in classes (default and with fewer tie bands), the bench line
out=$GRAFT_REPO_ROOT/gpurun_out/r06_run1
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -m gpu > $out/pytest_round6.log 2>&1; echo "pytest round6 rc=$?" | tee -a $out/summary.txt
tail -5 $out/pytest_round6.log | tee -a $out/summary.txt
for n in 1 2 4 8 16; do timeout 120 tests/cpp/test_threads --time $n 2 | tee -a $out/threads_default.jsonl; done
for b in 1 2 4; do for n in 1 4 16; do echo "bands $b" >> $out/threads_bands.jsonl; BRISK_TR_BANDS=$b timeout 120 tests/cpp/test_threads --time $n 2 | tee -a $out/threads_bands.jsonl; done; done
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" | tee -a $out/summary.txt
python tools/show_bench.py $out/bench.json 2>/dev/null | head -40
python - <<PY
import json
d=json.load(open("$out/bench.json"))
print(d["value"], d["roofline"]["frac"], json.dumps(d["config"]["pcie_fed"]))
print(json.dumps(d["config"]["other_configs"].get("threads"))[:1500])
PY
