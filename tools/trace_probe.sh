#!/bin/bash
# usage (GPU box): tools/trace_probe.sh <tag> <flags> : kernel trace of the two-context probe, overlap summary
tag=$1; flags=$2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export PROBE_NCTX=${PROBE_NCTX:-2} PROBE_CHUNKS=${PROBE_CHUNKS:-6}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out/kt -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/overlap_probe.py $flags > $out/probe.log 2>&1
tail -3 $out/probe.log
f=$(find $out/kt -name '*kernel_trace.csv' | head -1)
python3 $GRAFT_REPO_ROOT/tools/trace_overlap.py $f ${3:-60} | tee $out/overlap.txt
rm -rf $out/kt
