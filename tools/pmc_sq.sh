#!/bin/bash
# usage: tools/pmc_sq.sh <tag> [bench args]; SQ counter passes over a short bench run; raw csv + summary into gpurun_out/<tag>
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $out/p$i -o pass --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-fed --no-other-configs --batch 64 --inner 2 --steps 2 --warmup 1 "$@" > $out/p$i.log 2>&1
  tail -2 $out/p$i.log
done
find $out -name "*counter_collection.csv" | xargs python3 $GRAFT_REPO_ROOT/tools/pmc_sq.py --kernel k_detect | tee $out/summary_k_detect.txt
find $out -name "*counter_collection.csv" | xargs python3 $GRAFT_REPO_ROOT/tools/pmc_sq.py > $out/summary_all.txt
find $out -name "*counter_collection.csv" -size +2M -delete
