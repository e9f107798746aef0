#!/bin/bash
# PMC passes over conv_planar_kx3_kernel on the proto-net layer (scripts/bench_layers.py --set proto), shipped library and timing-ablation variants
# (stmask_amd/variants/libstmask_hip_<name>.so) -> gpurun_out/pmc_kx3/summary.txt.  rocprofv3 --pmc only with --kernel-trace.  usage: pmc_kx3.sh "default kabl2 kabl16"
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc_kx3; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for name in ${1:-default}; do
  if [ $name = default ]; then unset STM_LIBRARY; else export STM_LIBRARY=$R/stmask_amd/variants/libstmask_hip_$name.so; fi
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
             "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU" \
             "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum TCC_REQ_sum TCC_HIT_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/${name}_p$i -o p -- python3 $R/scripts/bench_layers.py --set proto > $OUT/${name}_p$i.log 2>&1
    echo "$name pass $i exit $?"
  done
done
cd $R
python3 - <<'PY' | tee $OUT/summary.txt
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_kx3"
tab = collections.defaultdict(dict)
for d in sorted(glob.glob(out + "/*_p*")):
    if not os.path.isdir(d): continue
    name = os.path.basename(d).rsplit("_p", 1)[0]
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "conv_planar_kx3" not in r.get("Kernel_Name", ""): continue
            k = r["Counter_Name"]; acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
        for k, (v, n) in acc.items():
            tab[k][name] = v / max(n, 1)
names = sorted({n for v in tab.values() for n in v})
print("%-34s" % "counter (per launch)" + "".join("%18s" % n for n in names))
for k in sorted(tab):
    print("%-34s" % k + "".join("%18.4g" % tab[k].get(n, float("nan")) for n in names))
PY
