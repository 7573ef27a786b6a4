#!/bin/bash
# k_wn_layer tuning probes on the GPU box: rebuild with -DFACPPG_ABLATE=n, 1 vs 2 workgroups per CU.
for a in ${ABLS:-0}; do
  make -s -C fac-via-ppg_amd/csrc clean >/dev/null; make -s -C fac-via-ppg_amd/csrc EXTRA=-DFACPPG_ABLATE=$a 2>/dev/null >/dev/null
  for lds in ${LDSS:-65536}; do
  FACPPG_WN_LDS=$lds FACPPG_BENCH_NO_CHECK=1 timeout 200 python bench.py --no-cpu-baseline --no-e2e --no-train --steps 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate=$a lds=$lds', 'layer_ms', round(d['roofline']['avg_launch_ms'],3), 'frac', round(d['roofline']['frac'],3))"
  done
done
