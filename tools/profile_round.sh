#!/bin/bash
# usage (on the GPU box): bash tools/profile_round.sh <tag> [--config <c> ...]
#   --config 4 | 4_uniform | 4_uniform_single | 5 | 1 | dense30 | dense50: afterwards the same per BASELINE configuration
#   (tools/profile_config.sh: entry of bench.py's config.other_configs, kernel statistics, FETCH / WRITE passes, traffic)
# Collects what profiles/ holds for a round into gpurun_out/<tag>/:
#   kernel_stats_b512.csv    rocprofv3 --kernel-trace --stats of the default bench command (512 frames per launch)
#   kernel_stats_b256.csv    the same at --batch 256
#   pmc_fetch_b64.csv / pmc_write_b64.csv   separate PMC passes (FETCH_SIZE, WRITE_SIZE) of a 64-frame run, engine kernels only
#   traffic.json             tools/pmc_traffic.py on those two passes (per kernel and per group, tagged with the kernel revision)
#   sq_counters.txt          SQ occupancy / stall / instruction counters per kernel (tools/pmc_sq.sh)
#   bench.json               the bench line of an un-profiled default run
tag=$1; shift
configs=""
while [ $# -gt 0 ]; do case $1 in --config) configs="$configs $2"; shift 2;; *) shift;; esac; done
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/bench.py > $out/bench.json 2> $out/bench.err
# the default command (512 frames per launch: what the bench line is quoted on) ...
timeout 300 rocprofv3 --kernel-trace --stats -d $out/ks5 -o r --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-fed --no-other-configs --steps 2 --warmup 1 --inner 4 > $out/ks5.log 2>&1
cp $(find $out/ks5 -name "*kernel_stats.csv" | head -1) $out/kernel_stats_b512.csv
rm -rf $out/ks5
# ... and 256 frames per launch (the unit of DESIGN.md's kernel tables and of the earlier rounds' files)
timeout 300 rocprofv3 --kernel-trace --stats -d $out/ks -o r --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-fed --no-other-configs --batch 256 --steps 2 --warmup 1 --inner 8 > $out/ks.log 2>&1
cp $(find $out/ks -name "*kernel_stats.csv" | head -1) $out/kernel_stats_b256.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c -d $out/pmc_$c -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-fed --no-other-configs --batch 64 --inner 2 --steps 2 --warmup 1 > $out/pmc_$c.log 2>&1
done
f=$(find $out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
w=$(find $out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
# keep the engine's kernels only (the raw files carry torch's set-up kernels with very long names)
python3 - "$f" "$out/pmc_fetch_b64.csv" "$w" "$out/pmc_write_b64.csv" <<'PY'
import csv, sys
for src, dst in ((sys.argv[1], sys.argv[2]), (sys.argv[3], sys.argv[4])):
    rows = list(csv.DictReader(open(src)))
    keep = [r for r in rows if r["Kernel_Name"].startswith("k_") or r["Kernel_Name"].startswith("void k_")]
    w = csv.DictWriter(open(dst, "w", newline=""), fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(keep)
PY
rev=$(python3 -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import ethzasl_brisk_amd as B; print(B.load_library().brisk_hip_kernel_revision().decode())")
python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py $out/pmc_fetch_b64.csv $out/pmc_write_b64.csv 64 1920 1080 $out/traffic.json $rev > /dev/null
cd $GRAFT_REPO_ROOT && bash tools/pmc_sq.sh $tag/sq > /dev/null 2>&1
cp $out/sq/summary_all.txt $out/sq_counters.txt
rm -rf $out/ks $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/sq; bash $GRAFT_REPO_ROOT/tools/pmc_describe.sh $tag/describe > /dev/null 2>&1; cp $out/describe/summary_describe.txt $out/describe_tcp_counters.txt
cat $out/bench.json | cut -c1-600
if [ -n "$configs" ]; then bash $GRAFT_REPO_ROOT/tools/profile_config.sh $tag $configs | tail -3; fi
