#!/bin/bash
# gpurun with retries when no GPU slot is free (exit code 3: nothing ran, nothing was charged).  usage: tools/gpurun_retry.sh <timeout> <command>
for attempt in 1 2 3 4 5 6 7 8; do
    /usr/local/graft/bin/gpurun --timeout $1 -- "$2"
    rc=$?
    if [ $rc -ne 3 ]; then exit $rc; fi
    sleep 150
done
exit 3
