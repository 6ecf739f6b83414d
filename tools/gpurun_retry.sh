#!/bin/bash
# gpurun_retry.sh <timeout-seconds> '<command>' : calls gpurun; when the pod has no free GPU slot / box (exit code 3: nothing
# ran, nothing was charged) waits and asks again, up to 12 times.  Any other outcome is returned as it is -- a command that ran is
# never repeated.
T=$1; shift
for i in $(seq 1 12); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 150
done
exit 3
