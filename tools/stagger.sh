#!/bin/bash
# Experiment: stagger the two co-resident k_wn_layer workgroups of a CU (build with -DFACPPG_STAGGER).
make -s -C fac-via-ppg_amd/csrc clean >/dev/null; make -s -j8 -C fac-via-ppg_amd/csrc EXTRA=-DFACPPG_STAGGER 2>/dev/null >/dev/null
for mode in ${MODES:-1 2 3}; do
for st in ${STAGGERS:-0 10 20 40}; do
  FACPPG_WN_STAGGER_MODE=$mode FACPPG_WN_STAGGER=$st timeout 200 python bench.py --no-cpu-baseline --no-e2e --no-train --steps 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mode=$mode stagger=$st', 'layer_ms', round(d['roofline']['avg_launch_ms'],3), 'frac', round(d['roofline']['frac'],4), 'ms/step', round(d['ms_per_step'],1))"
done
done
make -s -C fac-via-ppg_amd/csrc clean >/dev/null; make -s -j8 -C fac-via-ppg_amd/csrc 2>/dev/null >/dev/null
