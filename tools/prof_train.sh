#!/bin/bash
# rocprofv3 kernel-trace summary of the training step (tools/time_train.py <precision> <batch>) -> gpurun_out/prof_txt/
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_txt; mkdir -p $O; W=/tmp/facppg_prof_train; rm -rf $W; mkdir -p $W
for cfg in ${CFGS:-"bf16 3" "bf16 12"}; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --stats -d $W/$tag -o r -- python tools/time_train.py $cfg > $W/$tag.log 2>&1; echo "$tag rc=$?"
  grep seg= $W/$tag.log
  python tools/rocpd_summary.py stats $W/$tag/r_results.db | cut -c1-200 > $O/train_kernel_stats_$tag.txt
done
ls -la $O
