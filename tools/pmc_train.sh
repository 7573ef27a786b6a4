#!/bin/bash
# SQ / LDS / cache counters of the bf16 training step's kernels (two eager steps at batch $1, default 12; separate --pmc passes)
# -> gpurun_out/prof_txt/train_pmc_<set>_B<batch>.txt
export TMPDIR=/tmp
B=${1:-12}
O=$GRAFT_REPO_ROOT/gpurun_out/prof_txt; mkdir -p $O; W=/tmp/facppg_pmc_train; rm -rf $W; mkdir -p $W
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  STEPS=2 timeout 400 rocprofv3 --kernel-trace --pmc $set -d $W/$i -o r -- python tools/time_train_step.py $B > $W/$i.log 2>&1; echo "set $i rc=$? $(tail -1 $W/$i.log)"
  python tools/rocpd_summary.py pmc $W/$i/r_results.db "k_" | grep "k_bgemm\|k_wgrad\|k_wn_fwd\|k_wn_bwd\|k_dspect" | cut -c1-60,90-200 > $O/train_pmc_set${i}_B$B.txt
done
ls -la $O
