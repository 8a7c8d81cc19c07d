#!/bin/bash
# usage (GPU box): tools/pmc_describe.sh <tag> [bench args]: kernel trace + vector-memory path counters of the two k_describe stages
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $out/ks -o r --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-fed --no-other-configs --steps 2 --warmup 1 --inner 8 "$@" > $out/ks.log 2>&1
cp $(find $out/ks -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv; rm -rf $out/ks
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $out/p$i -o pass --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-fed --no-other-configs --batch 256 --inner 1 --steps 2 --warmup 1 "$@" > $out/p$i.log 2>&1
done
find $out -name "*counter_collection.csv" | xargs python3 $GRAFT_REPO_ROOT/tools/pmc_sq.py > $out/summary_all.txt
grep -A30 "^k_describe" $out/summary_all.txt > $out/summary_describe.txt
find $out -name "*counter_collection.csv" -delete
head -30 $out/kernel_stats.csv | cut -c1-150
cat $out/summary_describe.txt
