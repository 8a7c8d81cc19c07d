#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/icache
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $out/p1 -o pass --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-fed --no-other-configs --batch 64 --inner 2 --steps 2 --warmup 1 > $out/p1.log 2>&1
tail -3 $out/p1.log
timeout 300 rocprofv3 --pmc SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG -d $out/p2 -o pass --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-fed --no-other-configs --batch 64 --inner 2 --steps 2 --warmup 1 > $out/p2.log 2>&1
tail -3 $out/p2.log
find $out -name "*counter_collection.csv" | xargs python3 $GRAFT_REPO_ROOT/tools/pmc_sq.py > $out/summary_all.txt
find $out -name "*counter_collection.csv" -size +2M -delete
