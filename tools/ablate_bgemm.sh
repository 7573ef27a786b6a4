#!/bin/bash
# k_bgemm ablations (timing only; results are wrong): rebuild with -DFACPPG_BG_ABLATE=n and print the rocprof per-kernel averages
export TMPDIR=/tmp
for a in ${ABLS:-0 1 2 4 7}; do
  make -s -C fac-via-ppg_amd/csrc clean >/dev/null; make -s -j8 -C fac-via-ppg_amd/csrc EXTRA=-DFACPPG_BG_ABLATE=$a 2>/dev/null >/dev/null
  W=/tmp/abl_$a; rm -rf $W
  timeout 300 rocprofv3 --kernel-trace --stats -d $W -o r -- python tools/time_train.py bf16 12 > $W.log 2>&1
  echo "ablate=$a $(grep seg= $W.log | cut -c1-60)"
  python tools/rocpd_summary.py stats $W/r_results.db | grep "k_bgemm" | head -5 | awk '{print "   ", $3, $(NF-5), $(NF-3)}'
done
make -s -C fac-via-ppg_amd/csrc clean >/dev/null; make -s -j8 -C fac-via-ppg_amd/csrc 2>/dev/null >/dev/null
