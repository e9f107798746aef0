#!/bin/bash
# bench.py lines of the other BASELINE configurations (R50 FCB-ada, R101 FCB-ali at 8 / 32 clips, config 5 = R101 FCB-ali at 736x1280 with the fp16x1
# backbone and with fp16x2, the realistic regime at 8 clips) -> gpurun_out/<tag>/bench_configs.jsonl.  usage: scripts/bench_configs.sh [tag]
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r04}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
: > $OUT/bench_configs.jsonl
for cfg in STMask_plus_resnet50_ada_config STMask_plus_base_ali_config; do for clips in 8 32; do
  timeout 600 python bench.py --config $cfg --clips $clips --steps 16 --warmup 4 --no-cpu-baseline --no-extras >> $OUT/bench_configs.jsonl 2>> $OUT/bench_configs.err; echo "$cfg $clips exit $?"
done; done
timeout 600 python bench.py --config STMask_plus_base_ali_config --height 736 --width 1280 --planes fp16x1 --clips 4 --steps 16 --warmup 4 --no-cpu-baseline --no-extras >> $OUT/bench_configs.jsonl 2>> $OUT/bench_configs.err; echo "config5 fp16x1 exit $?"
timeout 600 python bench.py --config STMask_plus_base_ali_config --height 736 --width 1280 --clips 4 --steps 16 --warmup 4 --no-cpu-baseline --no-extras >> $OUT/bench_configs.jsonl 2>> $OUT/bench_configs.err; echo "config5 fp16x2 exit $?"
timeout 600 python bench.py --max-instances 8 --clips 8 --steps 16 --warmup 4 --no-cpu-baseline --no-extras >> $OUT/bench_configs.jsonl 2>> $OUT/bench_configs.err; echo "realistic 8 clips exit $?"
python - "$OUT/bench_configs.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d=json.loads(l); r=d.get('roofline',{})
    print(d['config']['workload'][:70], '|', d['config']['clips_per_gpu'], 'clips', d['value'], 'frames/s', d['ms_per_step'], 'ms', 'frac', r.get('frac'), 'issued', r.get('frac_issued'), 'trunk', r.get('frac_trunk_only'), 'tracked', d['config']['tracked_instances_mean'])
PY
