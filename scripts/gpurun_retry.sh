#!/bin/bash
# gpurun with patience: retries while the pod has no free GPU slot / box (exit code 3: nothing charged).
#   bash scripts/gpurun_retry.sh <timeout seconds> '<command>'
T=$1; shift
for i in $(seq 1 30); do
  gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
