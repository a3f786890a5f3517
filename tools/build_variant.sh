#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags]: builds ab/<name>.so (a variant of liblora_hip.so for tools/ab.sh) and prints the
# walker kernels' resource usage
name=$1; shift
mkdir -p ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -x hip -Wno-unused-function -mllvm -greedy-reverse-local-assignment -Rpass-analysis=kernel-resource-usage "$@" \
  -I gr_lora_amd/csrc -o ab/$name.so gr_lora_amd/csrc/lora_kernels.hip gr_lora_amd/csrc/lora_runtime.cpp gr_lora_amd/csrc/lora_channelizer.hip gr_lora_amd/csrc/lora_frame_check.cpp 2> ab/$name.log
grep -A12 "Function Name: .*walker[23]_kernel" ab/$name.log | grep -E "Function Name|VGPRs:|Spill|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* //' | paste - - - - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | sort -u
