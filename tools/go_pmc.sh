#!/bin/bash
# PMC passes for the wide tree kernel during the Go CNN rounds (tools/go_bench.py), each in its own rocprofv3 run:
#   tools/go_pmc.sh <tag> <go9|go19> [games]
TAG=${1:-r03}; GAME=${2:-go9}; GAMES=${3:-2048}
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
run_pass() {
  local name=$1; shift
  rm -rf /tmp/gopmc_$name
  timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-include-regex step_kernel_wide --output-format csv -d /tmp/gopmc_$name -o p -- \
      python3 $REPO/tools/go_bench.py --only $GAME --games9 $GAMES --games19 $GAMES --cnn-only --rounds 1600 --warm 3000 > $REPO/gpurun_out/${TAG}_${GAME}_pmc_${name}.log 2>&1 || return 1
  local f=$(find /tmp/gopmc_$name -name "*counter_collection.csv" | head -1)
  python3 $REPO/tools/pmc_summary.py "$f" step_kernel_wide > $REPO/gpurun_out/${TAG}_${GAME}_tree_pmc_${name}_summary.csv
  echo "pass $name done"
}
run_pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM && \
run_pass sq2 SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA
