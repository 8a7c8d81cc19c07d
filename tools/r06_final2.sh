#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06_final2
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?" | tee -a $out/summary.txt
grep -E "passed|failed" $out/pytest_gpu.log | tail -2 | tee -a $out/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $out/summary.txt
python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" | tee -a $out/summary.txt
python tools/show_bench.py $out/bench.json | tee -a $out/summary.txt
