#!/bin/bash
# prints VGPR / scratch / spill figures of the walker3 kernels (cross-compile, no GPU needed)
cd /root/repo && python gr_lora_amd/build.py -v 2>&1 | grep -E "error|Function Name|VGPRs:|ScratchSize|VGPRs Spill" | grep -A3 "error\|walker3\|demod_symbols_w3" | grep -v "^--" | sed 's/.*remark: *//' | sed 's/\[-Rpass[^]]*\]//g' | sed 's/_ZN8lora_hip//' | paste - - - - | cut -c1-170
