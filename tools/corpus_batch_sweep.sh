#!/bin/bash
# corpus workload (BASELINE config 4 on one GPU) at several synthesis batch sizes: ms per pass over ${UTTS:-384} utterances
for b in ${BATCHES:-16 32 48 64 96}; do
  python bench.py --workload corpus --utterances ${UTTS:-384} --corpus-batch $b --steps 2 --warmup 1 2>/dev/null | tail -1 > /tmp/cb.json
  python -c "import json; d=json.load(open('/tmp/cb.json')); print('batch $b: %.1f ms per pass, %.3f M samples/s' % (d['ms_per_step'], d['value'] / 1e6))"
done
