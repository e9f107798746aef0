# kernel traces (rocprofv3 --kernel-trace) of bench.py at 1 and 8 clips -> gpurun_out/trace_c1/stats_c{1,8}.md
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/trace_c1; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for c in 1 8; do
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/c$c -o b -- python3 $R/bench.py --clips $c --steps 60 --warmup 6 --no-cpu-baseline --no-extras --no-sampler-pass > $OUT/c$c.log 2>&1
t=$(ls $OUT/c$c/*/*kernel_trace.csv $OUT/c$c/*kernel_trace.csv 2>/dev/null | head -1); python3 $R/scripts/summarize_trace.py "$t" > $OUT/stats_c$c.md 2>&1; head -30 $OUT/stats_c$c.md | cut -c1-150
find $OUT/c$c -name '*kernel_trace.csv' -size +20M -delete
done
