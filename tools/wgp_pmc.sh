#!/bin/bash
# rocprofv3 counters of the persistent WaveGlow launch (k_wg_persist) on one 200-frame utterance; summaries -> gpurun_out/wgp_pmc/
export TMPDIR=/tmp FACPPG_POLL_LIMIT=5
O=$GRAFT_REPO_ROOT/gpurun_out/wgp_pmc; mkdir -p $O; W=/tmp/wgp_prof; rm -rf $W; mkdir -p $W
CMD="python tools/wgp_debug.py ${1:-200}"
timeout 200 rocprofv3 --kernel-trace --stats -d $W/stats -o r -- $CMD > $W/stats.log 2>&1; echo "stats rc=$?"
python tools/rocpd_summary.py stats $W/stats/r_results.db | cut -c1-180 | head -12 > $O/kernel_stats.txt
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $W/sq -o r -- $CMD > $W/sq.log 2>&1; echo "sq rc=$?"
python tools/rocpd_summary.py pmc $W/sq/r_results.db k_w | cut -c1-180 > $O/pmc_sq.txt
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS -d $W/sq2 -o r -- $CMD > $W/sq2.log 2>&1; echo "sq2 rc=$?"
python tools/rocpd_summary.py pmc $W/sq2/r_results.db k_w | cut -c1-180 > $O/pmc_sq2.txt
cat $O/kernel_stats.txt $O/pmc_sq.txt $O/pmc_sq2.txt
