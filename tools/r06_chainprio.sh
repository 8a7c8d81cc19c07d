#!/bin/bash
# round-5 review item 6 (i): the latency-bound chain beside the integral kernel (tie resolution, k_finalize, k_desc_prepare) at raised wave
# priority (s_setprio 2 / 3), A / B against the shipped library: bench line and per-kernel averages.  Variants are built on the box.
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
from ethzasl_brisk_amd import build
build.build_variant("libbrisk_chainprio2", ["BRISK_CHAIN_PRIO=2"])
build.build_variant("libbrisk_chainprio3", ["BRISK_CHAIN_PRIO=3"])
PY
out=gpurun_out/r06_chainprio; mkdir -p $out
for rep in 1 2; do
for lib in libbrisk_hip libbrisk_chainprio2 libbrisk_chainprio3; do
  BRISK_HIP_LIB=$GRAFT_REPO_ROOT/ethzasl_brisk_amd/$lib.so python3 bench.py --no-cpu-baseline --no-host-fed --no-other-configs > $out/$lib.$rep.json 2> $out/$lib.$rep.err
  python3 - $out/$lib.$rep.json $lib <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
st = d["config"]["stage_ms_per_chunk"]
print("%-22s %9.1f frames/s  chunk %.3f ms  tie %.3f finalize %.3f integral %.3f prepare %.3f describe %.3f" % (sys.argv[2], d["value"], d["config"]["ms_per_chunk"], st["k_tie_resolve"], st["k_finalize"], st["k_integral_final"], st["k_desc_prepare"], st["k_describe"]))
PY
done; done | tee $out/summary.txt
