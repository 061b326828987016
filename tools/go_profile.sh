#!/bin/bash
# kernel-trace statistics of the Go 9x9 CNN rounds (tools/go_bench.py), on the GPU box
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/kt_go
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_go -o k -- python3 $REPO/tools/go_bench.py --rounds 800 --only go9 > $REPO/gpurun_out/go_profile.log 2>&1
cp "$(find /tmp/kt_go -name '*kernel_stats.csv' | head -1)" $REPO/gpurun_out/r01k_go_kernel_stats.csv
