#!/bin/bash
# register / scratch / LDS use of every kernel of one translation unit:  tools/kres.sh viforsdes_amd/csrc/vsde_attn.hip [filter]
mkdir -p gpurun_out/tmp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -c "$1" -o gpurun_out/tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "error|Function Name|VGPRs:|AGPRs:|VGPRs Spill|ScratchSize|LDS Size" | sed 's/^.*remark: [^ ]* *//;s/\[-Rpass[^]]*\]//' \
  | awk '/Name:/{if (line) print line; line=$0; next} {line=line " | " $0} END{print line}' | sed 's/  */ /g' | grep -E "${2:-.}"
