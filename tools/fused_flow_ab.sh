#!/bin/bash
# timing experiments on k_wn_flow8 (FACPPG_WN_FUSED_DEBUG bits: 1 plain stores, 2 one launch per layer, 4 wait before the prologue,
# 8 no acquire behind the wait)
export FACPPG_WG_PERSIST=0 FACPPG_POLL_LIMIT=2
for d in ${DBGS:-0 2 6 10 14 4 8}; do for f in ${FUS:-1 2}; do echo "== FUSED=$f DEBUG=$d"; FACPPG_WN_FUSED_DEBUG=$d timeout 300 python tools/fused_flow_probe.py 200 2>&1 | grep "FUSED=$f\|equal"; done; done
