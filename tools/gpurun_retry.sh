#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3 = nothing charged).   tools/gpurun_retry.sh <timeout-s> '<command>'
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2" > /tmp/gpurun_try.log 2>&1
  rc=$?
  if ! grep -q "status=transient" /tmp/gpurun_try.log; then cat /tmp/gpurun_try.log; exit $rc; fi
  sleep 45
done
cat /tmp/gpurun_try.log; exit 3
