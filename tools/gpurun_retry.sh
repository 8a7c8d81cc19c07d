#!/bin/bash
# usage: tools/gpurun_retry.sh <timeout_s> '<command>'   - retries while no GPU slot is free (exit code 3)
t=$1; shift
for i in $(seq 1 60); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
