#!/bin/bash
# gpurun, re-submitted only while the pool answers "no free box / slot" (exit code 3: nothing ran, nothing was charged).
# Never used to retry a command that ran.  usage: tools/gpurun_wait.sh <timeout> '<command>'
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 150
done
exit 3
