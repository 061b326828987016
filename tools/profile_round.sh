#!/bin/bash
# Runs on the GPU box (via gpurun): (1) rocprofv3 --kernel-trace --stats of the default bench command, (2) PMC passes
# (each its own rocprofv3 run, counters never combined with trace domains) for one kernel of the same command.
# usage: tools/profile_round.sh <tag> <kernel-regex> [extra bench args]
# One population (--populations 1): per-kernel durations of kernels running alone, what DESIGN.md section 5 quotes.
set -o pipefail
TAG=$1; KREGEX=$2; shift; shift
EXTRA="$@"
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
rm -rf /tmp/kt_$TAG
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -o k -- \
    python3 $REPO/bench.py --no-cpu-baseline --no-secondary --populations 1 $EXTRA > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 1
cp "$(find /tmp/kt_$TAG -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_bench_kernel_stats.csv
echo "kernel stats done"
run_pass() {
  local name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout -k 10 600 rocprofv3 --pmc "$@" --kernel-include-regex "$KREGEX" --output-format csv -d /tmp/pmc_$name -o p -- \
      python3 $REPO/bench.py --no-cpu-baseline --no-secondary --no-profile --populations 1 $EXTRA > $OUT/${TAG}_pmc_${name}.log 2>&1 || return 1
  local f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  python3 $REPO/tools/pmc_summary.py "$f" "" > $OUT/${TAG}_pmc_${name}_summary.csv
  echo "pass $name done: $(wc -l < $OUT/${TAG}_pmc_${name}_summary.csv) lines"
}
run_pass fetch FETCH_SIZE && \
run_pass write WRITE_SIZE && \
run_pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
