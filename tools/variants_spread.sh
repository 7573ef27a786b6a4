#!/bin/bash
# A/B of the weight-load placement in k_wn_layer's K loop (-DFACPPG_WN_SPREAD=n) on the GPU box.
for v in ${VARS:-0 1 2 3 4}; do
  make -s -C fac-via-ppg_amd/csrc clean >/dev/null; make -s -j8 -C fac-via-ppg_amd/csrc EXTRA="-DFACPPG_WN_SPREAD=$v" 2>/dev/null >/dev/null
  for rep in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --no-e2e --no-train --steps 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('spread=$v', 'layer_ms', round(d['roofline']['avg_launch_ms'],3), 'frac', round(d['roofline']['frac'],4), 'ms/step', round(d['ms_per_step'],1))"
  done
done
make -s -C fac-via-ppg_amd/csrc clean >/dev/null; make -s -j8 -C fac-via-ppg_amd/csrc 2>/dev/null >/dev/null
