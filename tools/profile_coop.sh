#!/bin/bash
# rocprofv3 over the END-TO-END batch-1 step with the cooperating kernels launched as ordinary kernels (FACPPG_COOP_PLAIN=1: same
# kernels, same grids; rocprofv3 on this stack dies on hipLaunchCooperativeKernel).  Leaves text in gpurun_out/prof_txt/:
# per-kernel stats, the timeline of one step (which kernel ran next to which, on which queue), and SQ counters of the decoder /
# BiLSTM kernels (their own pass, --kernel-trace only, as the pool requires; PMC passes serialise the kernels: counters, not timing).
export TMPDIR=/tmp FACPPG_COOP_PLAIN=1
O=$GRAFT_REPO_ROOT/gpurun_out/prof_txt; mkdir -p $O; W=/tmp/facppg_prof_coop; rm -rf $W; mkdir -p $W
cd $GRAFT_REPO_ROOT
B="python bench.py --workload e2e --e2e-batch 1 --no-cpu-baseline --no-e2e --no-train"
timeout 300 rocprofv3 --kernel-trace --stats -d $W/stats -o r -- $B --steps 10 --warmup 3 > $W/stats.log 2>&1; echo "stats rc=$?"; tail -3 $W/stats.log
python tools/rocpd_summary.py stats $W/stats/r_results.db | cut -c1-190 > $O/e2e_kernel_stats_B1_T200.txt
python tools/rocpd_summary.py timeline $W/stats/r_results.db k_decoder -3 | cut -c1-150 > $O/e2e_timeline_B1_T200.txt
if [ -z "$NO_PMC" ]; then
# (counter passes serialise the kernels: the streamed path's frame collectors would wait for a decoder that is queued behind them)
export FACPPG_STREAM=0
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $W/sq -o r -- $B --steps 2 --warmup 1 > $W/sq.log 2>&1; echo "sq rc=$?"; tail -2 $W/sq.log
python tools/rocpd_summary.py pmc $W/sq/r_results.db k_decoder | cut -c1-190 > $O/e2e_pmc_sq_decoder_B1_T200.txt
python tools/rocpd_summary.py pmc $W/sq/r_results.db k_bilstm | cut -c1-190 >> $O/e2e_pmc_sq_decoder_B1_T200.txt
fi
ls -la $O
