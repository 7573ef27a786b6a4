#!/bin/bash
# rocprofv3 kernel-trace summaries of the whole PPG -> wav path (Tacotron encoder / BiLSTM / decoder / postnet, WaveGlow,
# denoiser, mel analysis) at batch 1 and at the 16-utterance ragged batch.  The cooperative launches (decoder, BiLSTM) are
# tried first; if rocprofv3 does not survive them the one-workgroup kernels are profiled instead.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_txt; mkdir -p $O
for cfg in "1 200" "16 400"; do
  tag=$(echo $cfg | tr ' ' '_')
  W=/tmp/pe_$tag; rm -rf $W
  if timeout 180 rocprofv3 --kernel-trace --stats -d $W -o r -- python tools/e2e_once.py $cfg > $W.log 2>&1 && [ -f $W/r_results.db ]; then
    echo "$tag: cooperative kernels profiled"
  else
    echo "$tag: rocprofv3 failed with cooperative launches (rc=$?); profiling FACPPG_DECODER_MODE=single FACPPG_BILSTM_MODE=single"
    tail -3 $W.log
    rm -rf $W
    FACPPG_DECODER_MODE=single FACPPG_BILSTM_MODE=single timeout 300 rocprofv3 --kernel-trace --stats -d $W -o r -- python tools/e2e_once.py $cfg > $W.log 2>&1
    tag=${tag}_single
  fi
  python tools/rocpd_summary.py stats $W/r_results.db | cut -c1-200 > $O/e2e_kernel_stats_$tag.txt
  head -30 $O/e2e_kernel_stats_$tag.txt | cut -c1-90,100-170
done
