#!/bin/bash
# kernel-trace statistics of the Go CNN rounds (tools/go_bench.py), on the GPU box: tools/go_profile.sh <tag> <go9|go19>
TAG=${1:-r02f}; GAME=${2:-go9}
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/kt_go
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_go -o k -- python3 $REPO/tools/go_bench.py --rounds 800 --only $GAME > $REPO/gpurun_out/${TAG}_${GAME}_profile.log 2>&1
cp "$(find /tmp/kt_go -name '*kernel_stats.csv' | head -1)" $REPO/gpurun_out/${TAG}_${GAME}_kernel_stats.csv
