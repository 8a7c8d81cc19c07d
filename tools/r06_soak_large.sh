#!/bin/bash
# the soak record of round 6 with larger case counts (the defaults of tools/soak.py finish in seconds on this engine)
out=$GRAFT_REPO_ROOT/gpurun_out/r06_soak_large
mkdir -p $out
cd $GRAFT_REPO_ROOT
rev=$(python3 -c "import ethzasl_brisk_amd as B; print(B.load_library().brisk_hip_kernel_revision().decode())")
echo "soak of kernel revision $rev ($(date -u +%Y-%m-%dT%H:%MZ)), tools/r06_soak_large.sh" > $out/summary.txt
run() { s=$1; shift; t0=$(date +%s); timeout 2400 python3 tools/soak.py $s "$@" > $out/$s.log 2>&1; rc=$?
  echo "suite $s $*: exit code $rc, $(( $(date +%s) - t0 )) s: $(grep -E "^$s:" $out/$s.log | tail -1)" | tee -a $out/summary.txt
  grep -E "MISMATCH|ERROR|HANG" $out/$s.log | head -5 | tee -a $out/summary.txt; }
run frames 160
run describe
run ordered
run callspace 6000 11
run options 6000 12
run matcher 3000 13
run large 1500 14
run threads 1500 1500
run hostpaths 1500 29
