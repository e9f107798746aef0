#!/bin/bash
# rocprofv3 kernel stats of the single-stream step (1 clip) and of 8 clips
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT/r03; cd /tmp; export TMPDIR=/tmp
for c in 1 8; do
  rm -rf $OUT/prof_c$c
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c$c -o bench -- python3 $R/bench.py --clips $c --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $OUT/prof_c$c.log 2>&1; echo "prof $c exit $?"
  tr=$(ls $OUT/prof_c$c/*kernel_trace.csv $OUT/prof_c$c/*/*kernel_trace.csv 2>/dev/null | head -1)
  (cd $R && python scripts/summarize_trace.py "$tr" 8 > $OUT/r03/kernel_stats_clips$c.md 2>&1)
  find $OUT/prof_c$c -name '*.csv' -size +5M -delete
done
head -70 $OUT/r03/kernel_stats_clips1.md | cut -c1-170
