#!/bin/bash
export TMPDIR=/tmp
run() {  # $1 label, env already set
  W=/tmp/var_$RANDOM; rm -rf $W
  timeout 300 rocprofv3 --kernel-trace --stats -d $W -o r -- python tools/time_train.py bf16 12 > $W.log 2>&1
  echo "[$1] $(grep seg= $W.log | cut -c1-75)"
  python tools/rocpd_summary.py stats $W/r_results.db 2>/dev/null | grep "k_bgemm\|k_wgrad(" | head -6 | awk '{print "      ", $3, $4, $(NF-3)}' | tr '\n' ';'; echo
}
IFS='|' read -ra VARS <<< "${VARIANTS:--DFACPPG_BG_NT=1|-DFACPPG_BG_NT=1 -DFACPPG_BG_PD=2|-DFACPPG_BG_NT=1 -DFACPPG_WG_NT=1|-DFACPPG_BG_NT=1 -DFACPPG_BG_XCD=0}"
for v in "${VARS[@]}"; do
  make -s -C fac-via-ppg_amd/csrc clean >/dev/null; make -s -j8 -C fac-via-ppg_amd/csrc EXTRA="$v" 2>/dev/null >/dev/null
  run "$v"
done
make -s -C fac-via-ppg_amd/csrc clean >/dev/null; make -s -j8 -C fac-via-ppg_amd/csrc 2>/dev/null >/dev/null
