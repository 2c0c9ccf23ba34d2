#!/bin/bash
# SQ counters of the chain's pass kernel (config 2, 3 candidates), one rocprofv3 --pmc pass per counter pair; run on the GPU
# box from the repository root (gpurun -- 'bash tools/pmc_pass_kernel.sh; python tools/read_pmc.py').
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS" "SQC_ICACHE_REQ SQC_ICACHE_MISSES" "SQC_ICACHE_HITS SQ_IFETCH" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" "SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F64" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64" "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32" "SQ_INSTS_VALU_MUL_F32 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM" "SQ_INSTS_VMEM SQ_INSTS_SMEM" "SQ_INSTS_FLAT SQ_INSTS_LDS_LOAD" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_EXP_GDS"; do
  n=$(echo $c | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_$n -o r -- python tools/profile_eval.py --cand 3 > gpurun_out/pmc_$n.log 2>&1 || echo "fail $n"
done
