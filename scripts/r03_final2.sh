#!/bin/bash
# round 3, refresh of every committed measurement after the chain kernel: driver bench line + layer table, rocprofv3 stats + PMC traffic, other configs, stage timings
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT/r03; cd $R
timeout 900 python bench.py --steps 20 --warmup 4 --layer-table > $OUT/r03/bench_n1.json 2> $OUT/r03/layer_table_32clips.txt; echo "bench exit $?"
bash scripts/r03_final.sh > $OUT/r03/final.log 2>&1; tail -3 $OUT/r03/final.log
bash scripts/r03_measure.sh configs stages > $OUT/r03/measure.log 2>&1; tail -12 $OUT/r03/measure.log
