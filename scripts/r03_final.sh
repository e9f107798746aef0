#!/bin/bash
# round 3, final GPU session: rocprofv3 kernel trace + stats of the driver's bench command, PMC traffic passes
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT/r03; cd $R
export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/prof.log 2>&1; echo "prof exit $?"
cd $R
tr=$(ls $OUT/prof/*kernel_trace.csv $OUT/prof/*/*kernel_trace.csv 2>/dev/null | head -1)
st=$(ls $OUT/prof/*kernel_stats.csv $OUT/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
python scripts/summarize_trace.py "$tr" 4 > $OUT/r03/bench_kernel_stats.md 2>&1; head -40 $OUT/r03/bench_kernel_stats.md
cp "$st" $OUT/r03/bench_kernel_stats.csv
find $OUT/prof -name '*kernel_trace.csv' -size +20M -delete
bash scripts/gpu_round.sh pmc > $OUT/r03/pmc.log 2>&1; tail -5 $OUT/r03/pmc.log
python scripts/make_pmc_json.py $OUT $OUT/r03/pmc_traffic.json 2>&1 | head -40
find $OUT/pmc_fetch $OUT/pmc_write -name '*.csv' -size +20M -delete
