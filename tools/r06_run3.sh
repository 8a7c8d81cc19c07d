#!/bin/bash
# round 6, third GPU pass: API / kernel statistics of the drop-in classes under 1, 8, 16 host threads (what saturates at ~9.5 k frames/s),
# the two-deep round pipeline in the sampling microbenchmark, TA / TCP busy counters of k_describe
out=$GRAFT_REPO_ROOT/gpurun_out/r06_run3
mkdir -p $out
cd $GRAFT_REPO_ROOT
bash tools/mb_il2.sh > $out/mb_il2.txt 2>&1; cp gpurun_out/il2/rates.json $out/mb_rates.json; grep -E "^all" $out/mb_il2.txt | head -30
T=$GRAFT_REPO_ROOT/tests/cpp/test_threads
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $out/list_avail.txt 2>&1
for n in 1 8 16; do
  timeout 300 rocprofv3 --kernel-trace --hip-runtime-trace --stats -d $out/thr$n -o r --output-format csv -- $T --time $n 1 > $out/thr$n.log 2>&1
  for f in kernel_stats hip_api_stats; do cp $(find $out/thr$n -name "*${f}.csv" | head -1) $out/thr${n}_$f.csv 2>/dev/null; done
  # concurrency from the kernel trace: busy time of the union of kernel intervals, mean kernels in flight
  python3 - $out/thr$n $n <<'PY' | tee -a $out/thr_summary.txt
import csv, glob, sys
d, n = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
t0, t1 = iv[len(iv)//4][0], iv[-len(iv)//4][1]   # the middle half of the run
iv = [(a, b) for a, b in iv if a >= t0 and b <= t1]
busy = 0; cur_a, cur_b = iv[0]
for a, b in iv[1:]:
    if a > cur_b: busy += cur_b - cur_a; cur_a, cur_b = a, b
    else: cur_b = max(cur_b, b)
busy += cur_b - cur_a
tot = sum(b - a for a, b in iv)
print("threads %s: kernels %d, window %.1f ms, union busy %.3f, mean kernels in flight while busy %.2f, kernel-seconds per second %.2f" % (n, len(iv), (t1 - t0) / 1e6, busy / (t1 - t0), tot / busy, tot / (t1 - t0)))
PY
  rm -rf $out/thr$n
done
cd $GRAFT_REPO_ROOT
for f in $out/thr*_hip_api_stats.csv $out/thr*_kernel_stats.csv; do echo "== $f"; head -14 $f | cut -c1-150; done > $out/thr_stats.txt
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_TA_TCP_STATE_READ_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  (cd /tmp; timeout 300 rocprofv3 --pmc $set -d $out/p$i -o pass --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-fed --no-other-configs --batch 256 --inner 1 --steps 2 --warmup 1 > $out/p$i.log 2>&1)
done
find $out -name "*counter_collection.csv" | xargs python3 tools/pmc_sq.py > $out/pmc_all.txt
grep -A40 "^k_describe" $out/pmc_all.txt | head -45 > $out/pmc_describe.txt
find $out -name "*counter_collection.csv" -delete; rm -rf $out/p?
cat $out/pmc_describe.txt
