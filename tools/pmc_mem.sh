#!/bin/bash
# usage: tools/pmc_mem.sh <tag> [kernel]; vector-memory path counters (TA / TCP / UTCL1) of a short bench run
tag=$1; kern=${2:-k_describe}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $out/p$i -o pass --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-fed --no-other-configs --batch 64 --inner 2 --steps 2 --warmup 1 > $out/p$i.log 2>&1
  tail -1 $out/p$i.log | cut -c1-200
done
find $out -name "*counter_collection.csv" | xargs python3 $GRAFT_REPO_ROOT/tools/pmc_sq.py --kernel $kern | tee $out/summary_$kern.txt
find $out -name "*counter_collection.csv" | xargs python3 $GRAFT_REPO_ROOT/tools/pmc_sq.py > $out/summary_all.txt
find $out -name "*counter_collection.csv" -delete
