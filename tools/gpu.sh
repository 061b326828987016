#!/bin/bash
# tools/gpu.sh [--timeout S] '<command>': rebuild every in-tree library (they travel with the snapshot), then run the command on
# the MI355X box through gpurun.  Keeps "the GPU ran a stale .so" from happening.
set -e
cd "$(dirname "$0")/.."
make -s -C sprl_amd/csrc
make -s -C oracle oracle
T=900
if [ "$1" = "--timeout" ]; then T=$2; shift 2; fi
exec /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
