#!/bin/bash
# usage (on the GPU box): bash tools/profile_config.sh <tag> <config> [<config> ...]
# config = an entry of bench.py's config.other_configs (1, 4, 4_uniform, 5, dense30, dense50).  Per config, into gpurun_out/<tag>/:
#   config<c>.json                 the entry (un-profiled run)
#   config<c>_kernel_stats.csv     rocprofv3 --kernel-trace --stats of `bench.py --config <c>`
#   config<c>_pmc_fetch.csv / _pmc_write.csv   separate PMC passes (FETCH_SIZE, WRITE_SIZE), engine kernels only
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --config $c > $out/config$c.json 2> $out/config$c.err
  timeout 240 rocprofv3 --kernel-trace --stats -d $out/ks_$c -o r --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --config $c --config-seconds 0.2 > $out/ks_$c.log 2>&1
  cp $(find $out/ks_$c -name "*kernel_stats.csv" | head -1) $out/config${c}_kernel_stats.csv
  if [ -z "$NO_PMC" ]; then
    for m in FETCH_SIZE WRITE_SIZE; do
      timeout 240 rocprofv3 --pmc $m -d $out/pmc_${c}_$m -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --config $c --config-seconds 0.05 > $out/pmc_${c}_$m.log 2>&1
      f=$(find $out/pmc_${c}_$m -name "*counter_collection.csv" | head -1)
      python3 - "$f" "$out/config${c}_pmc_$(echo $m | tr A-Z a-z | sed s/_size//).csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if r["Kernel_Name"].startswith("k_") or r["Kernel_Name"].startswith("void k_")]
w = csv.DictWriter(open(sys.argv[2], "w", newline=""), fieldnames=list(rows[0].keys()))
w.writeheader()
w.writerows(keep)
PY
      rm -rf $out/pmc_${c}_$m
    done
    # HBM bytes per launch and kernel (FETCH_SIZE x 2 + WRITE_SIZE, tools/pmc_traffic.py); frames per launch: 64 for the batches, 1 otherwise
    case $c in 4|4_uniform|dense30|dense50) fpl=64; geo="3840 2160";; *) fpl=1; geo="1920 1080";; esac
    case $c in dense30|dense50) geo="1920 1080";; 4_uniform_single) geo="3840 2160";; 1) geo="640 480";; esac
    rev=$(python3 -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import ethzasl_brisk_amd as B; print(B.load_library().brisk_hip_kernel_revision().decode())" 2>/dev/null)
    python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py $out/config${c}_pmc_fetch.csv $out/config${c}_pmc_write.csv $fpl $geo $out/config${c}_traffic.json $rev > /dev/null
  fi
  rm -rf $out/ks_$c
done
ls -la $out
