#!/bin/bash
# conv_chain.hip: tests + micro-benchmark
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_conv.py -q -x -k "bottleneck_chain" 2>&1 | tail -15
for b in 32 8 1; do timeout 300 python scripts/bench_chain.py $b 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03/chain.txt; done
