#!/bin/bash
# usage: gpurun_retry.sh <timeout> <command...>   -- retries while every GPU slot of the pod is busy (exit code 3)
T=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > /tmp/gpurun_last.txt 2>&1
  rc=$?
  if [ $rc -ne 3 ] && ! grep -q "status=transient" /tmp/gpurun_last.txt; then cat /tmp/gpurun_last.txt; exit $rc; fi
  sleep 45
done
cat /tmp/gpurun_last.txt; exit 3
