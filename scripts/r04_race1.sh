#!/bin/bash
# Round 4, GPU session 1: the co-residence race of the chain kernel -- reproduce, localise.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04; mkdir -p $OUT; cd $R
V=$R/stmask_amd/variants
echo "== 1. LDS-DMA ordering probe"; timeout 300 scripts/bin/ldsdma_order_probe 1500 > $OUT/ldsdma_order_probe.txt 2>&1; echo "exit $?"; cat $OUT/ldsdma_order_probe.txt | cut -c1-260
echo "== 2. pipeline-level reproducer: two ranks on one GPU, z-producing chain (STM_CHAIN_MODE=3)"
for i in 1; do STM_CHAIN_MODE=3 timeout 600 python bench.py --world2-one-gpu --clips 4 --steps 12 --warmup 3 > $OUT/world2_mode3_$i.json 2> $OUT/world2_mode3_$i.err; echo "exit $?"; python -c "
import json; d=json.load(open('$OUT/world2_mode3_$i.json')); print('mode3 run $i gather_ok', d['gather_ok'], 'max_abs', d['max_abs_diff_vs_solo'], [p['bit_equal_to_solo_run'] for p in d['per_rank']])"; done
echo "== 3. kernel-level stress, shipped library"
for h in pipe ew churn; do
  timeout 600 python scripts/ring_stress.py --hammer $h --launches 400 --cases chain --clips 4,32 --json $OUT/stress_$h.json > $OUT/stress_$h.txt 2>&1; echo "hammer $h exit $?"; grep -v "amdgpu.ids" $OUT/stress_$h.txt | cut -c1-330 | tail -60
done
echo "== 4. probe build beside the pipeline hammer (context switches / stalls)"
STM_LIBRARY=$V/libstmask_hip_probe.so timeout 600 python scripts/ring_stress.py --hammer pipe --probe --launches 160 --cases chain --clips 4 --json $OUT/stress_probe_pipe.json > $OUT/stress_probe_pipe.txt 2>&1; echo "exit $?"; grep -v "amdgpu.ids" $OUT/stress_probe_pipe.txt | cut -c1-330 | tail -40
STM_LIBRARY=$V/libstmask_hip_probe.so timeout 600 python scripts/ring_stress.py --hammer churn --probe --launches 160 --cases chain --clips 4 --json $OUT/stress_probe_churn.json > $OUT/stress_probe_churn.txt 2>&1; echo "exit $?"; grep -v "amdgpu.ids" $OUT/stress_probe_churn.txt | cut -c1-330 | tail -40
echo "== 5. counted-wait build (round 3's first producer loop)"
STM_LIBRARY=$V/libstmask_hip_counted.so timeout 600 python scripts/ring_stress.py --hammer pipe --launches 400 --cases chain --clips 4,32 --json $OUT/stress_counted_pipe.json > $OUT/stress_counted_pipe.txt 2>&1; echo "exit $?"; grep -v "amdgpu.ids" $OUT/stress_counted_pipe.txt | cut -c1-330 | tail -40
echo "== 6. kxr / planar rings beside the pipeline hammer"
timeout 600 python scripts/ring_stress.py --hammer pipe --launches 240 --cases kxr,planar --clips 32,4 --json $OUT/stress_conv_pipe.json > $OUT/stress_conv_pipe.txt 2>&1; echo "exit $?"; grep -v "amdgpu.ids" $OUT/stress_conv_pipe.txt | cut -c1-330 | tail -30
