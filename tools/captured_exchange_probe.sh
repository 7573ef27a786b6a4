#!/bin/bash
# Repeats the data-parallel training bench (RCCL, world 1, collectives captured in the step graph) with and without the
# settle time before the capture: how often does the c10d watchdog trip over an event of a stream that joined the capture?
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for settle in ${SETTLES:-400 0}; do
  ok=0; bad=0
  for i in 1 2 3 4 5 6; do
    if FACPPG_CAPTURE_SETTLE_MS=$settle MASTER_PORT=$((29600 + i)) timeout 300 python bench.py --force-dist --workload train --grad-dtype bf16 --steps 4 > gpurun_out/probe_${settle}_$i.log 2> gpurun_out/probe_${settle}_$i.err; then ok=$((ok+1)); else bad=$((bad+1)); fi
  done
  echo "settle ${settle} ms: $ok passed, $bad failed"
done
