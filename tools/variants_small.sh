#!/bin/bash
# A/B of build flags at batch 1 on the GPU box: VARIANTS="flags1|flags2|..." -> WaveGlow.infer ms at SHAPES (default 1x100 1x200)
IFS='|' read -ra VARS <<< "${VARIANTS:-|}"
for v in "${VARS[@]}"; do
  make -s -C fac-via-ppg_amd/csrc clean >/dev/null; make -s -j8 -C fac-via-ppg_amd/csrc EXTRA="$v" 2>/dev/null >/dev/null
  echo "[$v] $(timeout 200 python tools/time_wg.py ${SHAPES:-1x100 1x200} 2>/dev/null | tr '\n' '|')"
  echo "[$v] $(timeout 200 python tools/time_wg.py ${SHAPES:-1x100 1x200} 2>/dev/null | tr '\n' '|')"
done
make -s -C fac-via-ppg_amd/csrc clean >/dev/null; make -s -j8 -C fac-via-ppg_amd/csrc 2>/dev/null >/dev/null
