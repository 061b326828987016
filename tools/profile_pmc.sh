#!/bin/bash
# Runs on the GPU box (via gpurun): PMC passes for the tree kernel of the default bench command, each in its own
# rocprofv3 run (counters never combined with trace domains), summaries copied to gpurun_out/.
# usage: tools/profile_pmc.sh <tag> [extra bench args]
set -o pipefail
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
run_pass() {
  local name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout -k 10 500 rocprofv3 --pmc "$@" --kernel-include-regex step_kernel --output-format csv -d /tmp/pmc_$name -o p -- \
      python3 $REPO/bench.py --no-cpu-baseline --no-secondary --no-profile --populations 1 $EXTRA > $REPO/gpurun_out/${TAG}_pmc_${name}.log 2>&1 || return 1
  local f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  python3 $REPO/tools/pmc_summary.py "$f" step_kernel > $REPO/gpurun_out/${TAG}_pmc_${name}_summary.csv
  echo "pass $name done: $(wc -l < $REPO/gpurun_out/${TAG}_pmc_${name}_summary.csv) lines"
}
EXTRA="$@"
run_pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM && \
run_pass fetch FETCH_SIZE && \
run_pass write WRITE_SIZE && \
run_pass sq2 SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE
