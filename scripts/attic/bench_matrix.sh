#!/bin/bash
# The bench variants quoted in DESIGN.md section 6, one JSON line each (no CPU baseline).  Output: gpurun_out/bench_matrix.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/bench_matrix.txt; : > $OUT
run() { echo "== $*" >> $OUT; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | cut -c1-2400 >> $OUT; }
run
run --overlap early
run --overlap off
run --planes bf16x3
run --clips 8
run --clips 8 --overlap early
run --clips 8 --planes bf16x3
run --clips 8 --no-planar
run --clips 8 --config STMask_plus_resnet50_ada_config
run --clips 8 --config STMask_plus_base_ali_config
run --clips 4 --config STMask_plus_base_ali_config --height 736 --width 1280
run --clips 16
run --clips 64
grep -o '^== .*\|"value": [0-9.]*\|"achieved": [0-9.]*' $OUT | paste - - - - 2>/dev/null | cut -c1-200
