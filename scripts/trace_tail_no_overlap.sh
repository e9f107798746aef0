#!/bin/bash
# The step's tail kernels (everything that is not a convolution launch) WITHOUT the next trunk beside them: bench.py --overlap off, plain and under the
# rocprofv3 kernel trace -- what each tail kernel takes when it has the GPU to itself inside the step (the default step runs them beside the prefetched
# trunk on a second stream, which stretches their trace durations).   usage: gpurun -- 'bash scripts/trace_tail_no_overlap.sh'  -> gpurun_out/tail_no_overlap/
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/tail_no_overlap; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for ov in late off late off; do
  python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extras --no-sampler-pass --overlap $ov 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap $ov', d['value'], d['ms_per_step'], 'conv', d['roofline']['ms_per_step'])"
done | tee $OUT/step_times.txt
cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-sampler-pass --overlap off > $OUT/prof.log 2>&1
cd $R; t=$(ls $OUT/prof/*/*kernel_trace.csv $OUT/prof/*kernel_trace.csv 2>/dev/null | head -1); python scripts/summarize_trace.py "$t" > $OUT/kernel_stats_no_overlap.md 2>&1; sed -n 1,40p $OUT/kernel_stats_no_overlap.md | cut -c1-150
find $OUT/prof -name '*kernel_trace.csv' -size +20M -delete
