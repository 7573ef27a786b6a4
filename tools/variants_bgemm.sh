#!/bin/bash
# k_bgemm build variants x tile-width threshold: rocprof per-kernel averages of the bf16 training step at B = 12 and 3
export TMPDIR=/tmp
IFS='|' read -ra VARS <<< "${VARIANTS:--DFACPPG_BG_PD=2|-DFACPPG_BG_PD=2 -DFACPPG_BG_XCD=1|-DFACPPG_BG_PD=3|-DFACPPG_BG_PD=1}"
for v in "${VARS[@]}"; do
  make -s -C fac-via-ppg_amd/csrc clean >/dev/null; make -s -j8 -C fac-via-ppg_amd/csrc EXTRA="$v" 2>/dev/null >/dev/null
  for wide in ${WIDES:-384 100000}; do
   for B in 12 3; do
    W=/tmp/var_$RANDOM; rm -rf $W
    FACPPG_BG_WIDE=$wide timeout 300 rocprofv3 --kernel-trace --stats -d $W -o r -- python tools/time_train.py bf16 $B > $W.log 2>&1
    echo "[$v] wide>=$wide $(grep seg= $W.log | cut -c1-75)"
    python tools/rocpd_summary.py stats $W/r_results.db 2>/dev/null | grep "k_bgemm" | head -5 | awk '{print "      ", $3, $4, $(NF-3)}' | tr '\n' ';'; echo
   done
  done
done
make -s -C fac-via-ppg_amd/csrc clean >/dev/null; make -s -j8 -C fac-via-ppg_amd/csrc 2>/dev/null >/dev/null
