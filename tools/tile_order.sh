#!/bin/bash
# k_wn_layer tile order on an XCD (FACPPG_WN_TILE_ORDER): time (hipEvents in bench.py) and L2-miss traffic (PMC) per launch
export TMPDIR=/tmp
for o in ${ORDERS:-1 2 3}; do
  FACPPG_WN_TILE_ORDER=$o timeout 200 python bench.py --no-cpu-baseline --no-e2e --no-train --steps 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('order=$o layer_ms', round(d['roofline']['avg_launch_ms'],3), 'frac', round(d['roofline']['frac'],4))"
  for c in FETCH_SIZE WRITE_SIZE; do
    W=/tmp/to_${o}_$c; rm -rf $W
    FACPPG_WN_TILE_ORDER=$o timeout 300 rocprofv3 --kernel-trace --pmc $c -d $W -o r -- python bench.py --no-cpu-baseline --no-e2e --no-train --steps 1 --warmup 0 > $W.log 2>&1
    python tools/rocpd_summary.py pmc $W/r_results.db k_wn_layer | grep -v "^kernel" | awk -v c=$c -v o=$o '{print "   order=" o, c, $(NF-5), "calls", $(NF-4), "avg KiB", $(NF-3)}'
  done
done
