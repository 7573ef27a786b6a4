#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box; leaves only text summaries in gpurun_out/prof_txt/.
# (PMC passes are separate runs with --kernel-trace only, as the pool requires.)  Two shapes of the vocoder: BASELINE
# configs[1] (B = 8 x 1000 frames) and the metric's batch-1 utterance (B = 1 x 200: the streamed utterance's seed passes and mixed
# layer launches, tools/seeded_workload.py); the end-to-end step itself is traced by tools/profile_coop.sh (FACPPG_COOP_PLAIN=1).
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_txt; mkdir -p $O; W=/tmp/facppg_prof; rm -rf $W; mkdir -p $W
B="python bench.py --workload infer --no-cpu-baseline"
B1="python tools/seeded_workload.py"   # the headline's vocoder launches (k_cond_seed, k_wn_layer_mixed), see there
timeout 300 rocprofv3 --kernel-trace --stats -d $W/stats -o r -- $B --steps 3 --warmup 1 > $W/stats.log 2>&1; echo "stats rc=$?"
python tools/rocpd_summary.py stats $W/stats/r_results.db | cut -c1-190 > $O/kernel_stats.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $W/stats_b1 -o r -- $B1 10 > $W/stats_b1.log 2>&1; echo "stats b1 rc=$?"
python tools/rocpd_summary.py stats $W/stats_b1/r_results.db | cut -c1-190 > $O/kernel_stats_B1_T200.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $W/$c -o r -- $B --steps 1 --warmup 0 > $W/$c.log 2>&1; echo "$c rc=$?"
  python tools/rocpd_summary.py pmc $W/$c/r_results.db facppg | cut -c1-190 > $O/pmc_$c.txt
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $W/${c}_b1 -o r -- $B1 2 > $W/${c}_b1.log 2>&1; echo "$c b1 rc=$?"
  python tools/rocpd_summary.py pmc $W/${c}_b1/r_results.db facppg | cut -c1-190 > $O/pmc_${c}_B1_T200.txt
done
python tools/make_pmc_json.py $W/FETCH_SIZE/r_results.db $W/WRITE_SIZE/r_results.db $O/pmc.json "${PMC_NOTE:-}" $W/FETCH_SIZE_b1/r_results.db $W/WRITE_SIZE_b1/r_results.db > $O/pmc_json.log 2>&1; echo "pmc.json rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $W/sq -o r -- $B --steps 1 --warmup 0 > $W/sq.log 2>&1; echo "sq rc=$?"
python tools/rocpd_summary.py pmc $W/sq/r_results.db k_wn_layer | cut -c1-190 > $O/pmc_sq.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $W/sq_b1 -o r -- $B1 2 > $W/sq_b1.log 2>&1; echo "sq b1 rc=$?"
python tools/rocpd_summary.py pmc $W/sq_b1/r_results.db k_ | grep -E "k_wn_layer_mixed|k_cond_seed|^kernel" | cut -c1-190 > $O/pmc_sq_B1_T200.txt
ls -la $O
