#!/bin/bash
# tools/gpu_retry.sh [--timeout S] '<command>': tools/gpu.sh, retried ONLY while gpurun answers "no box or slot free" (exit 3:
# nothing ran, nothing was charged).  Any other exit code ends the loop: a command that ran is never run twice.
cd "$(dirname "$0")/.."
for i in $(seq 1 12); do
    tools/gpu.sh "$@"
    rc=$?
    if [ $rc -ne 3 ]; then exit $rc; fi
    echo "[gpu_retry] no slot free (attempt $i), waiting 120 s"
    sleep 120
done
exit 3
