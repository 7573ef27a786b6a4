#!/bin/bash
# k_wn_layer8 probes at batch 1 on the GPU box: rebuild with -DFACPPG_ABLATE8=n (1 no chunk barrier, 2 no activation loads,
# 4 weight stream from 8 KiB), time WaveGlow.infer at the shapes given (default 1x200).
for a in ${ABLS:-0 1 2 4 7}; do
  make -s -C fac-via-ppg_amd/csrc clean >/dev/null; make -s -j8 -C fac-via-ppg_amd/csrc EXTRA=-DFACPPG_ABLATE8=$a 2>/dev/null >/dev/null
  echo "ablate8=$a: $(timeout 200 python tools/time_wg.py ${SHAPES:-1x200} 2>/dev/null | tr '\n' '|')"
done
make -s -C fac-via-ppg_amd/csrc clean >/dev/null; make -s -j8 -C fac-via-ppg_amd/csrc 2>/dev/null >/dev/null
