#!/bin/bash
# One measurement session on a GPU box (what the committed profiles/rNN_* set is made from): the two hardware probes, the full GPU test suite, smoke,
# the driver's bench command (+ layer table), the rocprofv3 kernel trace + stats of it, and the PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate).
# usage: /usr/local/graft/bin/gpurun --timeout 5400 -- 'bash scripts/measure_round.sh r06 [first step, default 1; 3 = skip the probes and the test suite]'   -> gpurun_out/r06/*; copy what is to be judged into profiles/
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r06}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
FROM=${2:-1}
[ "$FROM" -le 2 ] && { echo "== 1. hardware probes: store data write-after-read per store form, LDS-DMA completion order"; timeout 600 scripts/bin/vmem_store_war_probe 300 > $OUT/store_war_probe_forms.txt 2>&1; echo "exit $?"; cut -c1-250 $OUT/store_war_probe_forms.txt | head -45; }
[ "$FROM" -le 2 ] && { timeout 300 scripts/bin/ldsdma_order_probe 1500 > $OUT/ldsdma_order_probe.txt 2>&1; echo "exit $?"; cut -c1-250 $OUT/ldsdma_order_probe.txt | head -10; }
[ "$FROM" -le 2 ] && { echo "== 2. pytest -m gpu"; timeout 2700 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu_final.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest_gpu_final.log; }
echo "== 3. smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $OUT/smoke.log
echo "== 4. bench (driver command) + layer table"; timeout 1500 python bench.py --steps 20 --warmup 4 --layer-table > $OUT/bench_final.json 2> $OUT/bench_final_layers.txt; echo "bench exit $?"
python - $OUT/bench_final.json <<'PY'
import json
import sys; d=json.load(open(sys.argv[1]))
r=d['roofline']
print(d['value'], d['ms_per_step'], 'frac', r['frac'], 'issued', r.get('frac_issued'), 'trunk', r.get('frac_trunk_only'), 'conv ms', r.get('ms_per_step'))
print('mfma', r['mfma_bound_launches']['frac'], r['mfma_bound_launches']['ms_per_step'], 'hbm', r['hbm_bound_launches']['frac'], r['hbm_bound_launches']['ms_per_step'])
print('im2col', d['roofline_im2col']['frac'], d['roofline_im2col']['avg_launch_us'])
f=d.get('roofline_dcn_fused') or {}
print('dcn_fused', f.get('frac'), f.get('avg_launch_us'), f.get('ms_per_step'), (f.get('hbm') or {}).get('frac'))
for k,v in d['extras'].items():
    if k == 'e2e':
        for kk, vv in v.items():
            if isinstance(vv, dict): print('e2e', kk, vv.get('value'), vv.get('ms_per_step'), vv.get('stages_ms_per_step'))
    else: print(k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'))
print(d['parity']['matched_frac'], d['parity']['mask_l2'], d['cpu_baseline']['value'])
PY
echo "== 5. rocprofv3 kernel trace + stats (--graph off --overlap late, like the pass bench.py records its per-launch events in: eager launches, no two convolution launches share the GPU)"
cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-sampler-pass --graph off --overlap late > $OUT/prof.log 2>&1; echo "prof exit $?"; cd $R
f=$(ls $OUT/prof/*/*kernel_stats.csv $OUT/prof/*kernel_stats.csv 2>/dev/null | head -1); echo "stats: $f"; head -12 "$f" | cut -c1-200
t=$(ls $OUT/prof/*/*kernel_trace.csv $OUT/prof/*kernel_trace.csv 2>/dev/null | head -1); python scripts/summarize_trace.py "$t" > $OUT/kernel_stats_final.md 2>&1; head -30 $OUT/kernel_stats_final.md | cut -c1-160
cp "$f" $OUT/kernel_stats_final.csv; find $OUT/prof -name '*kernel_trace.csv' -size +20M -delete
echo "== 6. PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes)"
rm -rf $OUT/pmc_fetch $OUT/pmc_write
cd /tmp
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --graph off > $OUT/pmc_fetch.log 2>&1; echo "pmc fetch exit $?"
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --graph off > $OUT/pmc_write.log 2>&1; echo "pmc write exit $?"
cd $R; python scripts/summarize_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt | cut -c1-200 | head -30
cp profiles/${TAG}_pmc_traffic.json $OUT/pmc_traffic.json 2>/dev/null; python scripts/make_pmc_json.py $OUT $OUT/pmc_traffic.json | head -40
find $OUT/pmc_fetch $OUT/pmc_write -name '*.csv' -size +8M -delete
