#!/bin/bash
# SQ stall breakdown of one conv shape (product kernel) next to the lab GEMM of the same M x K x N
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
CTR="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"
rm -rf /tmp/p1 /tmp/p2
rocprofv3 --kernel-trace --pmc $CTR -d /tmp/p1 -- python $R/scripts/conv_one.py $1 10 > /dev/null 2>&1
python $R/scripts/pmc_sq.py /tmp/p1 | head -12
rocprofv3 --kernel-trace --pmc $CTR -d /tmp/p2 -- python $R/scripts/dev/gemm_lab/run_one.py $2 > /dev/null 2>&1
python $R/scripts/pmc_sq.py /tmp/p2 | grep -A 11 "gemm_v" | head -12
