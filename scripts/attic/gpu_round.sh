#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof kernel trace.  Everything lands in gpurun_out/.
# usage: scripts/gpu_round.sh [tests] [smoke] [bench] [prof] [pmc]   (default: tests smoke bench prof)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
WHAT=${@:-tests smoke bench prof}
export TMPDIR=/tmp
for w in $WHAT; do
case $w in
tests)
  timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -25 $OUT/pytest_gpu.log ;;
smoke)
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log; tail -3 $OUT/smoke.log ;;
bench)
  timeout 900 python bench.py --steps 20 --warmup 4 > $OUT/bench.log 2> $OUT/bench.err; echo "bench exit $?"; tail -2 $OUT/bench.log; tail -3 $OUT/bench.err ;;
prof)
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/prof.log 2>&1
  echo "prof exit $?"; cd $R
  f=$(ls $OUT/prof/*/*kernel_stats.csv $OUT/prof/*kernel_stats.csv 2>/dev/null | head -1); echo "stats: $f"; head -25 "$f"
  # the raw per-dispatch trace is large: keep only the stats
  find $OUT/prof -name '*kernel_trace.csv' -size +20M -delete ;;
trunk)
  for m in "8 nobench" "8 bench" "8 bench cl"; do timeout 900 python scripts/bench_trunk.py $m >> $OUT/trunk.log 2>&1; done
  MIOPEN_FIND_MODE=FAST timeout 600 python scripts/bench_trunk.py 8 bench >> $OUT/trunk.log 2>&1
  grep -E "first pass|steady" $OUT/trunk.log ;;
kernels)
  timeout 1200 python scripts/bench_kernels.py --batch 8 > $OUT/kernels.log 2>&1; echo "kernels exit $?"; grep -E "TOTAL|gemm|corr|lincomb|nms|detect|roi|deform_conv" $OUT/kernels.log | head -70 ;;
pmcim2col)
  cd /tmp
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_im2col/p$i -o p -- python3 $R/scripts/prof_im2col.py 8 3 > $OUT/pmc_im2col_$i.log 2>&1; echo "pmc pass $i exit $?"
  done
  cd $R; python scripts/summarize_pmc_kernel.py $OUT/pmc_im2col deform_im2col_lds > $OUT/pmc_im2col_summary.txt 2>&1; head -30 $OUT/pmc_im2col_summary.txt
  find $OUT/pmc_im2col -name '*.csv' -size +5M -delete ;;
pmc)
  cd /tmp
  rm -rf $OUT/pmc_fetch $OUT/pmc_write
  timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $OUT/pmc_fetch.log 2>&1; echo "pmc fetch exit $?"
  timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $OUT/pmc_write.log 2>&1; echo "pmc write exit $?"
  cd $R; python scripts/summarize_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt ;;
esac
done
