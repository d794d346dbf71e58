#!/bin/bash
# gpurun with a wait for a free slot: retries ONLY on exit code 3 (no box / slot free: nothing ran, nothing charged).
#   tools/gpu.sh <timeout-seconds> '<command>'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
