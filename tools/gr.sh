#!/bin/bash
# development helper: gpurun with retries while no GPU slot is free.   tools/gr.sh LOGFILE TIMEOUT 'command'
log=$1; to=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@" > "$log" 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
