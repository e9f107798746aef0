#!/bin/bash
# PMC passes over scripts/bench_chain.py (rocprofv3 --pmc only with --kernel-trace; no other trace domains)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_chain
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_32B_sum TCC_WRITEBACK_sum" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 $R/scripts/bench_chain.py 32 > $OUT/p$i.log 2>&1
  echo "pass $i exit $?"
done
cd $R
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_chain"
for d in sorted(glob.glob(out + "/p*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            kn = r.get("Kernel_Name", "")
            key = "chain" if "conv_chain" in kn else "planar" if "conv_planar" in kn else None
            if key is None: continue
            k = key + " " + r["Counter_Name"]; acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
        for k, (v, n) in sorted(acc.items()):
            print(f"{os.path.basename(d)} {k:44s} per-launch {v / max(n, 1):16.1f}  (launches {n})")
PY
find $OUT -name '*.csv' -size +5M -delete
