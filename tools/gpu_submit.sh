#!/bin/bash
# development helper (authoring container only): submit one gpurun call, waiting while no GPU slot / box is free
# (exit code 3 = nothing charged).  usage: tools/gpu_submit.sh <timeout-seconds> '<command>'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
