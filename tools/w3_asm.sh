#!/bin/bash
# device-side ISA listing of lora_kernels.hip -> /tmp/w3/k.s, then the instruction mix of the named functions
mkdir -p /tmp/w3 && cd /root/repo/gr_lora_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip --cuda-device-only -S -o /tmp/w3/k.s lora_kernels.hip 2>&1 | grep -E "error" -A4 | head -20
cd /root/repo && python tools/asm_mix.py /tmp/w3/k.s "$@"
