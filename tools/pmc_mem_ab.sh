#!/bin/bash
# usage (GPU box): tools/pmc_mem_ab.sh <tag> <lib.so> [kernel]: vector-memory path counters of one kernel for a given build of the library
tag=$1; export BRISK_HIP_LIB=$2; kern=${3:-k_describe}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $out/p$i -o pass --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-fed --no-other-configs --batch 64 --inner 2 --steps 2 --warmup 1 > $out/p$i.log 2>&1
done
find $out -name "*counter_collection.csv" | xargs python3 $GRAFT_REPO_ROOT/tools/pmc_sq.py --kernel $kern | tee $out/summary_$kern.txt
find $out -name "*counter_collection.csv" -delete
