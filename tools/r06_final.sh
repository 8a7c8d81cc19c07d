#!/bin/bash
# round 6, final pass on the shipped revision: full GPU test suite, every soak suite, the round's profiles, the bench line
out=$GRAFT_REPO_ROOT/gpurun_out/r06_final
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?" | tee -a $out/summary.txt
tail -3 $out/pytest_gpu.log | tee -a $out/summary.txt
bash tools/r06_soak.sh
bash tools/profile_round.sh r06 > $out/profile_round.log 2>&1
python tools/show_bench.py gpurun_out/r06/bench.json | tee -a $out/summary.txt
python tools/probe_h2h.py 256 512 | tee $out/h2h.jsonl
