#!/bin/bash
# rocprofv3 kernel-trace summary of the graphed bf16 training step (tools/time_train_step.py <batch>, 44 steps: 2 eager warm-ups, the
# capture, 41 replays -- the one-off launches of the warm-ups, e.g. the optimiser-state initialisation, stay in the totals)
# -> gpurun_out/prof_txt/train_graph_kernel_stats_bf16_B<batch>.txt
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_txt; mkdir -p $O; W=/tmp/facppg_prof_train; rm -rf $W; mkdir -p $W
for B in ${BATCHES:-3 12}; do
  STEPS=44 timeout 300 rocprofv3 --kernel-trace --stats -d $W/$B -o r -- python tools/time_train_step.py $B > $W/$B.log 2>&1; echo "B=$B rc=$? $(tail -1 $W/$B.log)"
  python tools/rocpd_summary.py stats $W/$B/r_results.db | cut -c1-200 > $O/train_graph_kernel_stats_bf16_B$B.txt
done
ls -la $O
