#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 --kernel-trace --stats of the DEFAULT bench command (two populations; without the
# extra one-population pass, so that every convolution launch in the statistics is a half-size launch of the timed kind).
# usage: tools/profile_default.sh <tag>
set -o pipefail
TAG=$1
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
rm -rf /tmp/kt_$TAG
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -o k -- \
    python3 $REPO/bench.py --no-cpu-baseline --no-secondary --no-alone-pass > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 1
cp "$(find /tmp/kt_$TAG -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_bench_kernel_stats.csv
echo "kernel stats done"
