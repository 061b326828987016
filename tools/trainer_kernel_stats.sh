#!/bin/bash
# Runs on the GPU box: rocprofv3 --kernel-trace --stats of one epoch of the trainer at the BASELINE shape (eager steps: the per-kernel
# view of an optimiser step).  usage: tools/trainer_kernel_stats.sh <tag>
set -o pipefail
TAG=$1
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
rm -rf /tmp/kt_tr_$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_tr_$TAG -o k -- \
    python3 $REPO/tools/trainer_profile.py --samples 131072 --epochs 1 --eager > $OUT/${TAG}_trainer_profile.txt 2> $OUT/${TAG}_trainer_profile.err || exit 1
cp "$(find /tmp/kt_tr_$TAG -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_trainer_kernel_stats.csv
head -25 $OUT/${TAG}_trainer_kernel_stats.csv | cut -c1-200
