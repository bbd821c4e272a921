#!/bin/bash
# local helper (runs in the build container): gpurun with retries while every GPU slot of the pod is busy (exit code 3)
#   tools/r04/gpu.sh <timeout_s> '<command>'
t=$1; shift
for i in 1 2 3 4 5 6 7 8 9 10; do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
