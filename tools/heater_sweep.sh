#!/bin/bash
# decoder heaters: how many, how hard, from when?  whole end-to-end batch-1 step (tools/host_overhead_probe.py) per setting
run() { r=$(env "$@" python tools/host_overhead_probe.py 2>&1 | grep "tacotron.inference\|whole step" | sed 's/  */ /g; s/(.*)//' | tr '\n' '|'); echo "$*: $r"; }
run FACPPG_DECODER_HEATERS=0
run FACPPG_DECODER_HEATERS=-1
for lead in 30 60 100 150; do run FACPPG_DECODER_HEATERS=-1 FACPPG_DECODER_HEAT_LEAD=$lead; done
run FACPPG_DECODER_HEATERS=0
run FACPPG_DECODER_HEATERS=-1
