#!/bin/bash
# round 6, second GPU pass: what limits the drop-in classes under host threads (hardware queues, pageable uploads, the second upload),
# and where the host-to-host batch rate loses against the H2D-only rate
out=$GRAFT_REPO_ROOT/gpurun_out/r06_run2
mkdir -p $out
cd $GRAFT_REPO_ROOT
T=tests/cpp/test_threads
for n in 1 4 8 16; do timeout 120 $T --time $n 2 | tee -a $out/threads.jsonl; done
echo "--- GPU_MAX_HW_QUEUES=16" | tee -a $out/threads.jsonl
for n in 4 8 16; do GPU_MAX_HW_QUEUES=16 timeout 120 $T --time $n 2 | tee -a $out/threads.jsonl; done
echo "--- pinned" | tee -a $out/threads.jsonl
for n in 1 4 8 16; do timeout 120 $T --time $n 2 --pinned | tee -a $out/threads.jsonl; done
echo "--- pinned + same image" | tee -a $out/threads.jsonl
for n in 1 4 8 16; do timeout 120 $T --time $n 2 --pinned --same-image | tee -a $out/threads.jsonl; done
echo "--- pinned + same image + GPU_MAX_HW_QUEUES=16" | tee -a $out/threads.jsonl
for n in 4 8 16 32; do GPU_MAX_HW_QUEUES=16 timeout 120 $T --time $n 2 --pinned --same-image | tee -a $out/threads.jsonl; done
echo "--- VGA pinned + same image + GPU_MAX_HW_QUEUES=16" | tee -a $out/threads.jsonl
for n in 1 16; do GPU_MAX_HW_QUEUES=16 timeout 120 $T --time $n 2 640 480 --pinned --same-image | tee -a $out/threads.jsonl; done
python tools/probe_h2h.py 64 256 512 | tee $out/h2h.jsonl
echo "--- egress off" | tee -a $out/h2h.jsonl
BRISK_EXPORT_EGRESS=0 python tools/probe_h2h.py 256 | tee -a $out/h2h.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $out/prof -o r --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/probe_h2h.py 256 > $out/prof.log 2>&1
for f in $(find $out/prof -name "*stats.csv"); do echo "== $f"; head -12 $f | cut -c1-200; done | tee $out/prof_stats.txt
rm -rf $out/prof
