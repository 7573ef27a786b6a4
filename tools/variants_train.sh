#!/bin/bash
# A/B of build flags on the graphed bf16 training step: VARIANTS="flags1|flags2|..." -> tools/graph_train.py 3 12
IFS='|' read -ra VARS <<< "${VARIANTS:-|}"
for v in "${VARS[@]}"; do
  make -s -C fac-via-ppg_amd/csrc clean >/dev/null; make -s -j8 -C fac-via-ppg_amd/csrc EXTRA="$v" 2>/dev/null >/dev/null
  echo "[$v] $(timeout 200 python tools/graph_train.py 3 12 2>/dev/null | grep 'bf16 B=' | tr '\n' '|')"
done
make -s -C fac-via-ppg_amd/csrc clean >/dev/null; make -s -j8 -C fac-via-ppg_amd/csrc 2>/dev/null >/dev/null
