#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box; leaves only text summaries in gpurun_out/prof_txt/.
# (PMC passes are separate runs with --kernel-trace only, as the pool requires.)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_txt; mkdir -p $O; W=/tmp/facppg_prof; rm -rf $W; mkdir -p $W
B="python bench.py --no-cpu-baseline --no-e2e --no-train"
timeout 300 rocprofv3 --kernel-trace --stats -d $W/stats -o r -- $B --steps 3 --warmup 1 > $W/stats.log 2>&1; echo "stats rc=$?"
python tools/rocpd_summary.py stats $W/stats/r_results.db | cut -c1-190 > $O/kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $W/$c -o r -- $B --steps 1 --warmup 0 > $W/$c.log 2>&1; echo "$c rc=$?"
  python tools/rocpd_summary.py pmc $W/$c/r_results.db facppg | cut -c1-190 > $O/pmc_$c.txt
done
python tools/make_pmc_json.py $W/FETCH_SIZE/r_results.db $W/WRITE_SIZE/r_results.db $O/pmc.json "${PMC_NOTE:-}" > $O/pmc_json.log 2>&1; echo "pmc.json rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $W/sq -o r -- $B --steps 1 --warmup 0 > $W/sq.log 2>&1; echo "sq rc=$?"
python tools/rocpd_summary.py pmc $W/sq/r_results.db k_wn_layer | cut -c1-190 > $O/pmc_sq.txt
ls -la $O
