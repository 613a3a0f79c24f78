#!/bin/bash
# Round 5, fourth GPU call: what bounds the one-frame propagation launches -- PMC counters of the split-operand deformable conv and
# of the one-frame split-operand Winograd kernel (rocprofv3 --pmc, one pass per group, kernel-trace only).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
G="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU/SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE/SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE/TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum"
bash tools/pmc_raw.sh prop "$G" python $REPO/tools/prop_one.py 30 all 2>&1 | tail -120
