#!/bin/bash
# round 3 measurement deliverables: other configs, stage timings, clock probe, PMC of one layer, kernel microbench
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
for w in ${@:-configs stages kernels clock pmclayer}; do
case $w in
configs)
  : > $OUT/bench_configs.jsonl
  for cfg in STMask_plus_resnet50_ada_config STMask_plus_base_ali_config; do for clips in 8 32; do
    timeout 600 python bench.py --config $cfg --clips $clips --steps 16 --warmup 4 --no-cpu-baseline --no-extras >> $OUT/bench_configs.jsonl 2>> $OUT/bench_configs.err; echo "$cfg $clips exit $?"
  done; done
  timeout 600 python bench.py --config STMask_plus_base_ali_config --height 736 --width 1280 --planes fp16x1 --clips 4 --steps 16 --warmup 4 --no-cpu-baseline --no-extras >> $OUT/bench_configs.jsonl 2>> $OUT/bench_configs.err; echo "config5 fp16x1 exit $?"
  timeout 600 python bench.py --config STMask_plus_base_ali_config --height 736 --width 1280 --clips 4 --steps 16 --warmup 4 --no-cpu-baseline --no-extras >> $OUT/bench_configs.jsonl 2>> $OUT/bench_configs.err; echo "config5 fp16x2 exit $?"
  timeout 600 python bench.py --max-instances 8 --clips 8 --steps 16 --warmup 4 --no-cpu-baseline --no-extras >> $OUT/bench_configs.jsonl 2>> $OUT/bench_configs.err; echo "realistic 8 clips exit $?"
  python - <<'PY'
import json
for l in open('gpurun_out/r03/bench_configs.jsonl'):
    d=json.loads(l); r=d.get('roofline',{})
    print(d['config']['workload'][:70], '|', d['config']['clips_per_gpu'], 'clips', d['value'], 'frames/s', d['ms_per_step'], 'ms', 'frac', r.get('frac'), 'trunk', r.get('frac_trunk_only'), 'tracked', d['config']['tracked_instances_mean'])
PY
  ;;
stages) timeout 600 python scripts/bench_stages.py > $OUT/stage_timings.txt 2>&1; grep -v amdgpu.ids $OUT/stage_timings.txt ;;
kernels) timeout 1200 python scripts/bench_kernels.py --batch 32 > $OUT/kernel_microbench.txt 2>&1; echo "kernels exit $?"; grep -v amdgpu.ids $OUT/kernel_microbench.txt | tail -40 ;;
clock)
  (cd stmask_amd/csrc && touch conv_bf16x.hip && make -s EXTRA=-DSTM_ABLATE 2>&1 | grep error)
  { echo "# rocm-smi while the 256x128 fp16x2 ring kernel (256->256 3x3 at 96x160, batch 32) runs back to back; STM_CONV_ABL as in conv_bf16x.hip";
    for abl in 0 7 1 32 64; do echo "== STM_CONV_ABL=$abl (0 full loop, 7 MFMAs alone: no DMA / barrier / fragment reads, 1 no DMA, 32 no activation DMA, 64 no weight DMA)"; STM_CONV_ABL=$abl timeout 120 python scripts/clock_probe.py conv16 2>&1 | grep -v amdgpu.ids; done;
    echo "== elementwise (x.mul_)"; timeout 120 python scripts/clock_probe.py ew 2>&1 | grep -v amdgpu.ids; } > $OUT/clock_probe.txt 2>&1
  (cd stmask_amd/csrc && touch conv_bf16x.hip && make -s 2>&1 | grep error)
  cat $OUT/clock_probe.txt ;;
pmclayer) bash scripts/pmc_layer.sh > $OUT/pmc_layer.txt 2>&1; tail -45 $OUT/pmc_layer.txt ;;
esac
done
