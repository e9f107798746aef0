#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/prof1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof1 -o b1 -- python3 $R/bench.py --clips ${1:-1} --steps 60 --warmup 8 --no-cpu-baseline --no-extras > $OUT/prof1.log 2>&1
cd $R
tr=$(ls $OUT/prof1/*kernel_trace.csv $OUT/prof1/*/*kernel_trace.csv 2>/dev/null | head -1)
python scripts/summarize_trace.py "$tr" 10 2>&1 | head -60 | cut -c1-170
tail -1 $OUT/prof1.log | cut -c1-200
