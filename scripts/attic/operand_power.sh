#!/bin/bash
# Reproduces profiles/r04_operand_power.txt: the timing builds of conv_planar_kx3_kernel (RESULTS OF THESE BUILDS ARE WRONG BY CONSTRUCTION) on the proto-net
# layer.  Build the variants on the build host first:   for n in 1 65 2 66 32 16; do make -C stmask_amd/csrc variant NAME=kabl$n VSRC=conv_bf16x VFLAGS=-DKX3_ABL=$n; done
# then on a GPU box:   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash scripts/operand_power.sh'
cd ${GRAFT_REPO_ROOT:-$(pwd)}
bash scripts/variants_layers.sh "default kabl1 kabl65 kabl2 kabl66 kabl32 kabl16" --set proto 2>&1 | grep -E "^==|proto"
