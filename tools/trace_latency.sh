#!/bin/bash
# usage (GPU box): tools/trace_latency.sh <tag> [vga|1080p]: HIP API + kernel timeline of one detect() + compute() pair
tag=$1; which=${2:-vga}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/probe_latency.py $which 300 > $out/untraced.txt 2>&1
rocprofv3 --kernel-trace --hip-runtime-trace --memory-copy-trace -d $out/kt -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/probe_latency.py $which 30 > $out/probe.log 2>&1
tail -2 $out/probe.log
a=$(find $out/kt -name '*hip_api_trace.csv' | head -1)
k=$(find $out/kt -name '*kernel_trace.csv' | head -1)
m=$(find $out/kt -name '*memory_copy_trace.csv' | head -1)
python3 $GRAFT_REPO_ROOT/tools/trace_latency.py $a $k $m > $out/timeline.txt
rm -rf $out/kt
cat $out/untraced.txt
