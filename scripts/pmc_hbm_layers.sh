#!/bin/bash
# HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x 2 as on gfx950) of each HBM-bound layer shape of scripts/bench_layers.py --set hbm,
# launch by launch: are the 1x1 layers' inputs fetched once, or once per channel tile?   usage: gpurun -- 'bash scripts/pmc_hbm_layers.sh' -> gpurun_out/pmc_hbm_layers/
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc_hbm_layers; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
python3 $R/scripts/bench_layers.py --set hbm 2>&1 | grep -v amdgpu.ids > $OUT/plain.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o p -- python3 $R/scripts/bench_layers.py --set hbm > $OUT/$c.log 2>&1; echo "$c exit $?"
done
cd $R; python3 - <<'PY'
import csv, glob, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_hbm_layers"
names = [l[:52].strip() for l in open(out + "/plain.txt") if " us " in l and "TF" in l]
algo = [l.split("(")[-1].split(" MB")[0] for l in open(out + "/plain.txt") if " us " in l and "TF" in l]
us = [l.split(" us")[0].split()[-1] for l in open(out + "/plain.txt") if " us " in l and "TF" in l]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(out + f"/{c}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c and any(k in r["Kernel_Name"] for k in ("conv_planar_k", "conv_kxr_kernel", "conv_planar_kx3"))]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    vals = [float(r["Counter_Value"]) for r in rows]
    kn = [r["Kernel_Name"][:60] for r in rows]
    res[c] = (vals, kn)
n = len(names)
per = len(res["FETCH_SIZE"][0]) // max(n, 1)
print(f"{n} layers, {len(res['FETCH_SIZE'][0])} conv dispatches -> {per} per layer")
for i, nm in enumerate(names):
    fv = res["FETCH_SIZE"][0][i * per + 3:(i + 1) * per]; wv = res["WRITE_SIZE"][0][i * per + 3:(i + 1) * per]
    # FETCH_SIZE / WRITE_SIZE count 64-byte units... the guide's gfx950 rule: KB units, FETCH x 2
    f_mb = sum(fv) / max(len(fv), 1) * 1024 * 2 / 1e6; w_mb = sum(wv) / max(len(wv), 1) * 1024 / 1e6
    print(f"{nm:52s} {us[i]:>8s} us  algorithmic {algo[i]:>5s} MB | fetched {f_mb:7.1f} MB  written {w_mb:7.1f} MB  | {res['FETCH_SIZE'][1][i * per + 3]}")
PY
