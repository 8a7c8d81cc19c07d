#!/bin/bash
# usage (GPU box): tools/ab_kernel.sh <kernel name pattern> <lib.so> ... : per-kernel average of a short default bench run per library variant
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/ab_$(basename $lib .so)
  mkdir -p $out
  BRISK_HIP_LIB=$GRAFT_REPO_ROOT/$lib rocprofv3 --kernel-trace --stats -d $out -o r --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-fed --no-other-configs --steps 5 --warmup 2 $AB_BENCH_ARGS > $out/run.log 2>&1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "== $lib: $(tail -1 $out/run.log | cut -c1-60)"
  grep -E "$pat" $f | awk -F, '{printf "   %-50s calls %s avg %.1f us\n", substr($1,1,50), $2, $4/1000}'
  find $out -name "*kernel_trace.csv" -delete
done
