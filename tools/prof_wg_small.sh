#!/bin/bash
# rocprofv3 kernel-trace summary of WaveGlow.infer at small shapes (tools/time_wg.py BxT ...) -> gpurun_out/prof_txt/wg_small_<tag>.txt
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_txt; mkdir -p $O; W=/tmp/facppg_prof_small; rm -rf $W; mkdir -p $W
tag=${TAG:-small}
timeout 300 rocprofv3 --kernel-trace --stats -d $W/$tag -o r -- python tools/time_wg.py "$@" > $W/$tag.log 2>&1; echo "rc=$?"
cat $W/$tag.log | tail -3
python tools/rocpd_summary.py stats $W/$tag/r_results.db | cut -c1-200 > $O/wg_small_$tag.txt
head -12 $O/wg_small_$tag.txt | cut -c1-90,100-175
