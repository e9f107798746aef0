#!/bin/bash
# Timing ablations of dcn_fused_kernel (RESULTS OF THE ABLATED BUILDS ARE WRONG): build with
#   make -C stmask_amd/csrc variant NAME=abl<n> VSRC=dcn_fused VFLAGS=-DDF_ABL=<n>
#     (1 no corner gathers, 2 no blend / split / staging, 4 no MFMAs, 8 no weight DMA, 16 no fragment reads; sums combine)
#   make -C stmask_amd/csrc variant NAME=psched0 VSRC=dcn_fused VFLAGS=-DDF_PSCHED=0      (the compiler's order in the producer stream)
# then on the GPU box: bash scripts/abl_dcn_fused.sh [batch] [layer index]: one line per layer and build.
export LAYER=${2:-}
for v in "" $(ls stmask_amd/variants/ 2>/dev/null | sed -n 's/^libstmask_hip_\(.*\)\.so$/\1/p'); do
  if [ -z "$v" ]; then echo "== shipped"; FUSED_ONLY=1 python scripts/bench_dcn_fused.py ${1:-32};
  else echo "== $v"; FUSED_ONLY=1 STM_LIBRARY=$PWD/stmask_amd/variants/libstmask_hip_$v.so python scripts/bench_dcn_fused.py ${1:-32}; fi
done 2>&1 | grep -v "amdgpu.ids"
