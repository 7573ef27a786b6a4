#!/bin/bash
# decoder heaters: how many, how hard?  whole end-to-end batch-1 step (tools/host_overhead_probe.py) per setting
for hh in 0 -1 96; do for sl in 0 16 64 200; do
  [ "$hh" = "0" ] && [ "$sl" != "0" ] && continue
  r=$(FACPPG_DECODER_HEATERS=$hh FACPPG_DECODER_HEAT_SLEEP=$sl python tools/host_overhead_probe.py 2>&1 | grep "tacotron.inference\|whole step" | sed 's/  */ /g' | tr '\n' '|')
  echo "heaters $hh sleep $sl: $r"
done; done
