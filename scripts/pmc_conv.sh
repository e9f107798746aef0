#!/bin/bash
# PMC passes (counters only, with --kernel-trace) on the planar split conv kernel.
# usage: scripts/pmc_conv.sh [tag] [prof_conv.py args...]   e.g.  STM_CONV_DEBUG=5 scripts/pmc_conv.sh dbg5 2 2 1 big
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
TAG=${1:-conv}; shift
ARGS=${@:-2}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf $OUT/pmc_$TAG/p$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$TAG/p$i -o p -- python3 $R/scripts/prof_conv.py $ARGS > $OUT/pmc_${TAG}_$i.log 2>&1; echo "pass $i exit $?"; tail -1 $OUT/pmc_${TAG}_$i.log | cut -c1-200
done
cd $R; python3 scripts/summarize_pmc_kernel.py $OUT/pmc_$TAG conv_planar | cut -c1-600
find $OUT/pmc_$TAG -name '*.csv' -size +5M -delete
