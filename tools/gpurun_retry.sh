#!/bin/bash
# gpurun with patience: re-submits while the pod's GPU slots are all busy (exit code 3: nothing ran, nothing was charged).
# usage: tools/gpurun_retry.sh <timeout seconds> '<command>'
T=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
