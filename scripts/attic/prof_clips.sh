# rocprofv3 kernel trace of bench.py at a given clip count -> gpurun_out/prof_clipsN/kernel_stats.md   usage: prof_clips.sh N [steps]
R=${GRAFT_REPO_ROOT:-$(pwd)}; N=${1:-8}; S=${2:-30}; OUT=$R/gpurun_out/prof_clips$N; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 $R/bench.py --clips $N --steps $S --warmup 6 --no-cpu-baseline --no-extras > $OUT/prof.log 2>&1; echo "prof exit $?"; cd $R
t=$(ls $OUT/prof/*/*kernel_trace.csv $OUT/prof/*kernel_trace.csv 2>/dev/null | head -1); python scripts/summarize_trace.py "$t" > $OUT/kernel_stats.md 2>&1; head -64 $OUT/kernel_stats.md | cut -c1-170
find $OUT/prof -name '*kernel_trace.csv' -size +20M -delete
