#!/bin/bash
# A/B of k_wn_layer build flags on the GPU box: VARIANTS="flags1|flags2|..." (each: rebuild with EXTRA=flags, bench 3 steps twice)
IFS='|' read -ra VARS <<< "${VARIANTS:-}"
for v in "${VARS[@]}"; do
  make -s -C fac-via-ppg_amd/csrc clean >/dev/null; make -s -j8 -C fac-via-ppg_amd/csrc EXTRA="$v" 2>/dev/null >/dev/null
  for rep in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --no-e2e --no-train --steps 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v]', 'layer_ms', round(d['roofline']['avg_launch_ms'],3), 'frac', round(d['roofline']['frac'],4), 'ms/step', round(d['ms_per_step'],1))"
  done
done
make -s -C fac-via-ppg_amd/csrc clean >/dev/null; make -s -j8 -C fac-via-ppg_amd/csrc 2>/dev/null >/dev/null
