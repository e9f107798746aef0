#!/bin/bash
# true per-dispatch kernel durations (rocprofv3 kernel trace) for the im2col variants and ablations at batch 8
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
MB="207.8,160.6,103.1,79.5,51.3,39.5"
run() { # name variant debug
  rm -rf $OUT/abl_$1
  STM_IM2COL_DEBUG=$3 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/abl_$1 -o t -- python3 $R/scripts/prof_im2col.py 8 5 $2 > $OUT/abl_$1.log 2>&1
  echo "== $1 (variant $2, debug $3)"; python3 $R/scripts/trace_durations.py $OUT/abl_$1 deform_im2col $MB
}
run v2 2 0
run v3 3 0
run v3_storeonly 3 1
run v3_nostore 3 2
