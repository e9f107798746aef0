#!/bin/bash
# PMC passes over dcn_fused_kernel on one DCN layer shape (scripts/bench_dcn_fused.py FUSED_ONLY=1 LAYER=<index>; rocprofv3 --pmc only with --kernel-trace)
# usage (GPU box): bash scripts/pmc_dcn_fused.sh [layer index 0..6, default 1] [batch, default 32]
R=${GRAFT_REPO_ROOT:-$(pwd)}
L=${1:-1}; B=${2:-32}
OUT=$R/gpurun_out/pmc_dcn_fused
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export FUSED_ONLY=1 LAYER=$L
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_READ_LDS_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 $R/scripts/bench_dcn_fused.py $B > $OUT/p$i.log 2>&1
  echo "pass $i exit $?"
done
cd $R
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_dcn_fused"
for d in sorted(glob.glob(out + "/p*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "dcn_fused" not in r.get("Kernel_Name", ""): continue
            k = r["Counter_Name"]; acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
        for k, (v, n) in sorted(acc.items()):
            print(f"{os.path.basename(d)} {k:40s} per-launch {v / max(n, 1):18.1f}  (launches {n})")
PY
