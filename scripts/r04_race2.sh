#!/bin/bash
# Round 4, GPU session 2: the store write-after-read exposure -- hardware probe, the fixes, the drained-barrier form of the ring.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04; mkdir -p $OUT; cd $R
V=$R/stmask_amd/variants
echo "== 1. store WAR probe, alone on the GPU"; timeout 300 scripts/bin/vmem_store_war_probe 400 > $OUT/store_war_probe_solo.txt 2>&1; echo "exit $?"; cut -c1-330 $OUT/store_war_probe_solo.txt
echo "== 1b. store WAR probe beside the pipeline hammer"
python scripts/gpu_hammer.py pipe 45 > /dev/null 2>&1 &
HP=$!; sleep 30
timeout 300 scripts/bin/vmem_store_war_probe 400 > $OUT/store_war_probe_pipe.txt 2>&1; echo "exit $?"; cut -c1-330 $OUT/store_war_probe_pipe.txt | head -20
wait $HP
run() { # name library launches cases
  STM_LIBRARY=$2 timeout 900 python scripts/ring_stress.py --hammer pipe --launches $3 --cases chain --clips 4,32 --json $OUT/stress2_$1.json > $OUT/stress2_$1.txt 2>&1; echo "$1 exit $?"
  grep -v "amdgpu.ids" $OUT/stress2_$1.txt | grep -v "^      pixel\|^    output\|^  launch" | cut -c1-200
}
echo "== 2. round-3 code (no wait states behind the stores): baseline"; run r03 $V/libstmask_hip_r03.so 800
echo "== 3. wait states pinned behind every store (the shipped library)"; run fixnop $R/stmask_amd/libstmask_hip.so 2400
echo "== 4. slab offset in the VGPR offset, soffset 0 (the compiler pads)"; run fixvoff $V/libstmask_hip_fixvoff.so 1600
echo "== 5. drained barriers only, stores as in round 3"; run lgkm $V/libstmask_hip_lgkm.so 800
echo "== 6. counted producer wait + store fix"; run counted $V/libstmask_hip_counted.so 1200
echo "== 7. counted producer wait + store fix + drained consumer barriers"; run counted_lgkm $V/libstmask_hip_counted_lgkm.so 1200
echo "== 8. shipped producer loop + store fix + drained consumer barriers"; run fix_lgkm $V/libstmask_hip_fix_lgkm.so 1200
