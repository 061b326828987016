#!/bin/bash
# Runs on the GPU box: rocprofv3 --kernel-trace --stats of eager optimiser steps of the trainer at the BASELINE shape (the per-kernel
# view of a step).  The same command runs once BEFORE the traced run so that the library's solver search (cached on
# disk per user) is not in the trace.  usage: tools/trainer_kernel_stats.sh <tag> [extra trainer_profile.py flags, e.g. --no-fast-conv]
set -o pipefail
TAG=$1; shift
EXTRA="$@"
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
python3 $REPO/tools/trainer_profile.py --samples 131072 --epochs 1 --eager $EXTRA > $OUT/${TAG}_trainer_untraced.txt 2>&1 || exit 1
rm -rf /tmp/kt_tr_$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_tr_$TAG -o k -- \
    python3 $REPO/tools/trainer_profile.py --samples 131072 --epochs 1 --eager $EXTRA > $OUT/${TAG}_trainer_profile.txt 2> $OUT/${TAG}_trainer_profile.err || exit 1
cp "$(find /tmp/kt_tr_$TAG -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_trainer_kernel_stats.csv
tail -1 $OUT/${TAG}_trainer_untraced.txt; tail -1 $OUT/${TAG}_trainer_profile.txt
