#!/bin/bash
# gpurun, retried only while the pool answers "no slot / no box free" (exit 3: nothing ran, nothing charged).
# usage: tools/gpurun_wait.sh TIMEOUT 'command'
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
