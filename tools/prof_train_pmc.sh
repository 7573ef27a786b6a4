#!/bin/bash
# PMC pass (kernel-trace + counters only) over the bf16 training step: MFMA busy, LDS bank conflicts, per kernel
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_txt; mkdir -p $O; W=/tmp/facppg_prof_train_pmc; rm -rf $W; mkdir -p $W
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $W/sq -o r -- python tools/time_train.py bf16 12 > $W/sq.log 2>&1; echo "sq rc=$?"
python tools/rocpd_summary.py pmc $W/sq/r_results.db facppg | grep "k_bgemm\|k_wgrad(\|kernel " | cut -c1-200 > $O/train_pmc_sq_bf16_12.txt
head -40 $O/train_pmc_sq_bf16_12.txt | cut -c1-60,88-190
