#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 --kernel-trace of ONE step of the default bench command (two populations), then
# tools/timeline.py over the per-launch trace: how the two streams' kernels share the GPU (overlap, idle time, launch gaps).
# usage: tools/profile_timeline.sh <tag> [extra bench args]
set -o pipefail
TAG=$1; shift
EXTRA="$@"
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
rm -rf /tmp/tl_$TAG
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tl_$TAG -o k -- \
    python3 $REPO/bench.py --steps 1 --no-cpu-baseline --no-secondary --no-alone-pass $EXTRA > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 1
cp "$(find /tmp/tl_$TAG -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_bench_kernel_stats.csv
python3 $REPO/tools/timeline.py "$(find /tmp/tl_$TAG -name '*kernel_trace.csv' | head -1)" --skip 0.1 --json $OUT/${TAG}_timeline.json > $OUT/${TAG}_timeline.txt
cat $OUT/${TAG}_timeline.txt
