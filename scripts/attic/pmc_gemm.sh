#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*|SQ_INSTS_[A-Z_0-9]*|SQ_WAIT[A-Z_0-9_]*|SQ_ACTIVE_INST[A-Z_0-9_]*|SQ_LDS[A-Z_0-9_]*)\b" | sort -u | tr '\n' ' ' > $OUT/pmc_names.txt; cat $OUT/pmc_names.txt; echo
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SMEM"; do
  i=$((i+1)); rm -rf $OUT/pmc_gemm/p$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_gemm/p$i -o p -- python3 $R/scripts/prof_gemm.py 3 > $OUT/pmc_gemm_$i.log 2>&1; echo "pass $i exit $?"; tail -2 $OUT/pmc_gemm_$i.log | cut -c1-200
done
cd $R; python3 scripts/summarize_pmc_kernel.py $OUT/pmc_gemm gemm128 | cut -c1-400
