#!/bin/bash
# Runs on the GPU box (via gpurun): memory-path PMC passes for the 8x8 trunk convolution alone (tools/wino_lab, product kernel),
# each in its own rocprofv3 run (counters never combined with trace domains), summaries copied to gpurun_out/.
# usage: tools/conv_pmc.sh <tag> [boards]
set -o pipefail
TAG=$1; B=${2:-13492}
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
run_pass() {
  local name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout -k 5 90 rocprofv3 --pmc "$@" --kernel-include-regex "wino_conv64_kernel" --output-format csv -d /tmp/pmc_$name -o p -- \
      $REPO/tools/wino_lab $B 1 > $REPO/gpurun_out/${TAG}_pmc_${name}.log 2>&1 || return 1
  local f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  python3 $REPO/tools/pmc_summary.py "$f" "wino_conv64_kernel" > $REPO/gpurun_out/${TAG}_pmc_${name}_summary.csv
  echo "pass $name done: $(wc -l < $REPO/gpurun_out/${TAG}_pmc_${name}_summary.csv) lines"
}
# (at most two counters of a TA / TCP / TCC block per pass: more "exceeds the capabilities of the hardware")
run_pass sq SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY && \
run_pass sq3 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS && \
run_pass ta1 TA_BUSY_avr GRBM_GUI_ACTIVE && \
run_pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum && \
run_pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum && \
run_pass tcp2 TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum && \
run_pass tcp3 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum && \
run_pass tcp4 TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum && \
run_pass tcc1 TCC_HIT_sum TCC_MISS_sum && \
run_pass tcc2 TCC_BUSY_avr TCC_TAG_STALL_sum
