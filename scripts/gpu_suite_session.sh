# the whole GPU test suite + smoke + the default bench line (what the driver runs at round end), outputs under gpurun_out/suite/
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/suite; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench.json')); r=d['roofline']
print('value',d['value'],'ms',d['ms_per_step'],'frac',r['frac'],'issued',r.get('frac_issued'),'trunk',r.get('frac_trunk_only'),'mfma',r['mfma_bound_launches']['frac'],r['mfma_bound_launches']['ms_per_step'],'hbm',r['hbm_bound_launches']['frac'],r['hbm_bound_launches']['ms_per_step'],'im2col',d['roofline_im2col']['frac'],'clips1',d['extras']['clips1']['value'],'clips8',d['extras']['clips8']['value'],'realistic',d['extras']['realistic']['value'],'parity',d['parity']['matched'],d['parity']['mask_l2'])"
