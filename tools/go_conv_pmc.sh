#!/bin/bash
# HBM traffic of the any-board trunk convolution during Go CNN rounds (tools/go_bench.py, one population, bounded rounds): FETCH_SIZE
# and WRITE_SIZE in separate rocprofv3 --pmc passes, then profiles-ready JSON (what bench.py's Go lines read as roofline.traffic):
#   tools/go_conv_pmc.sh <tag> <go9|go19> [games]
TAG=${1:-r04}; GAME=${2:-go9}; GAMES=${3:-2048}
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
run_pass() {
  local name=$1; shift
  rm -rf /tmp/gocpmc_$name
  timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-include-regex wino_conv64_nchw --output-format csv -d /tmp/gocpmc_$name -o p -- \
      python3 $REPO/tools/go_bench.py --only $GAME --games9 $GAMES --games19 $GAMES --cnn-only --rounds 1200 --warm 400 > $REPO/gpurun_out/${TAG}_${GAME}_conv_pmc_${name}.log 2>&1 || return 1
  local f=$(find /tmp/gocpmc_$name -name "*counter_collection.csv" | head -1)
  python3 $REPO/tools/pmc_summary.py "$f" wino_conv64_nchw > $REPO/gpurun_out/${TAG}_${GAME}_conv_pmc_${name}_summary.csv
  echo "pass $name done"
}
run_pass fetch FETCH_SIZE && run_pass write WRITE_SIZE && python3 $REPO/tools/go_traffic_json.py $TAG $GAME
