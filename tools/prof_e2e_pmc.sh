#!/bin/bash
# rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) over the whole PPG -> wav path at the
# 16-utterance ragged batch, one-workgroup decoder / BiLSTM kernels (rocprofv3 does not survive the cooperative launches).
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_txt; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  W=/tmp/pe_pmc_$c; rm -rf $W
  FACPPG_DECODER_MODE=single FACPPG_BILSTM_MODE=single timeout 300 rocprofv3 --kernel-trace --pmc $c -d $W -o r -- python tools/e2e_once.py 16 400 > $W.log 2>&1; echo "$c rc=$?"
  python tools/rocpd_summary.py pmc $W/r_results.db "" | cut -c1-200 > $O/e2e_pmc_${c}_16_400_single.txt
  head -25 $O/e2e_pmc_${c}_16_400_single.txt | cut -c1-90,100-190
done
