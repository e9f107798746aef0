#!/bin/bash
# true per-dispatch durations (rocprofv3 kernel trace) of im2col configurations at batch 8
# usage: exp_im2col.sh "tag|variant|ENV1=.. ENV2=.." ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for spec in "$@"; do
  tag=${spec%%|*}; rest=${spec#*|}; var=${rest%%|*}; envs=${rest#*|}
  rm -rf $OUT/exp_$tag
  env $envs timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/exp_$tag -o t -- python3 $R/scripts/prof_im2col.py 8 5 $var > $OUT/exp_$tag.log 2>&1
  python3 - "$OUT/exp_$tag" "$tag [$envs]" <<'PY'
import csv,glob,sys
MB=[207.8,160.6,103.1,79.5,79.5,51.3,39.5]; names=["L1.0","L1.2","L2.0","L2.2","L2.4","L3.0","L3.2"]
fs=glob.glob(sys.argv[1]+"/*kernel_trace.csv")
if not fs: print(sys.argv[2],"NO TRACE"); sys.exit()
rows=[r for r in csv.DictReader(open(fs[0])) if "deform_im2col" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
tot=0; line=""
for i,n in enumerate(names):
    d=sorted((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows[i*5:(i+1)*5])
    if not d: continue
    tot+=d[len(d)//2]; line+=f" {n}:{d[len(d)//2]:.1f}/{MB[i]/d[len(d)//2]*1e3:.0f}"
print(f"{sys.argv[2]:60s} TOTAL {tot:6.1f} us {sum(MB)/tot*1e3:5.0f} GB/s |{line}")
PY
done
