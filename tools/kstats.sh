#!/bin/bash
# usage: tools/kstats.sh <tag> [bench args] : rocprofv3 kernel-trace stats of a short bench run -> gpurun_out/<tag>/
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out -o r --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" > $out/run.log 2>&1
tail -1 $out/run.log | cut -c1-300
f=$(find $out -name "*kernel_stats.csv" | head -1)
column -s, -t < $f | cut -c1-150 | head -20
find $out -name "*kernel_trace.csv" -delete
