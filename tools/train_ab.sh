#!/bin/bash
# Same-box A/B of the bf16 training step: the round-5 launch structure (default) against its switches turned off -- two-launch layers
# forward and backward, 128 x 128 weight-gradient tiles in launch order, k_bgemm for the conditioning gradient -- interleaved, three
# times.  (The switched-off runs still contain the round's arithmetic changes: v_cvt_pk_bf16_f32 packing, v_exp / v_rcp gate, weight
# norm and start-conv kernels; against round 4's build the difference is larger.)
for rep in 1 2 3; do
  echo "round-5 structure:  $(timeout 200 python tools/time_train_step.py ${BATCHES:-3 12})"
  echo "switches off:       $(FACPPG_TRAIN_FUSED_FWD=0 FACPPG_TRAIN_FUSED_BWD=0 FACPPG_WGRAD_TILE=128 FACPPG_TRAIN_DSPECT_BGEMM=1 FACPPG_WGRAD_NO_XCD_MAP=1 timeout 200 python tools/time_train_step.py ${BATCHES:-3 12})"
done
