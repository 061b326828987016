#!/bin/bash
# registers / spills / scratch of the 8x8 trunk-convolution instantiations of cnn_wino.hip:  tools/kres8.sh [extra hipcc flags]
cd "$(dirname "$0")/../sprl_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -c -o /tmp/kres8.o cnn_wino.hip -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
  awk '/error/ {print} /Function Name:/ {n=$0; sub(/.*wino_conv64_kernelILi/, "", n); sub(/EEEv.*/, "", n); name=n} /VGPRs:/ {v=$0; sub(/.*VGPRs: /, "", v); sub(/ .*/, "", v)} /ScratchSize/ {s=$0; sub(/.*: /, "", s); sub(/ .*/, "", s)} /VGPRs Spill/ {p=$0; sub(/.*Spill: /, "", p); sub(/ .*/, "", p); if (name ~ /^8ELi8E/) print "H,W,HEADS,RES,PERSIST,NPRE = " name "  VGPRs " v "  scratch " s " B  spilled " p}'
