#!/bin/bash
# usage: tools/flag_sweep.sh <flags...> : bench with each --debug-flags value (timing experiments; results are wrong on purpose)
for f in "$@"; do bash tools/bench_sweep.sh "--steps 10 --debug-flags $f"; done
