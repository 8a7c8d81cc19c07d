#!/bin/bash
# usage (GPU box): tools/sweep_env.sh <tag> VAR v1 v2 ... : bench stage times for each value of an engine tuning knob
tag=$1; var=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
for v in "$@"; do
  env $var=$v python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-fed --no-other-configs --steps 3 --warmup 1 --inner 8 > $out/$var.$v.json 2> $out/$var.$v.err
  python3 - $out/$var.$v.json $var $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], sys.argv[3], d["value"], d["config"]["ms_per_chunk"], d["config"]["stage_ms_per_chunk"])
PY
done
