#!/bin/bash
# Turn gpurun_out/ of tools/refresh_profiles.sh into the committed summaries under profiles/ (round tag = $1, default r01).
R=${1:-r06}
O=gpurun_out
P=profiles
cp $O/prof_stats/run_kernel_stats.csv $P/${R}_bench_kernel_stats.csv
grep "^{" $O/bench_line.json > $P/${R}_bench_line.json
[ -f $O/bench_line_dist1.json ] && grep "^{" $O/bench_line_dist1.json > $P/${R}_bench_line_dist1.json
[ -f $O/timeline.txt ] && cp $O/timeline.txt $P/${R}_bench_step_timeline.txt
[ -f $O/timeline_fixed_plan.txt ] && cp $O/timeline_fixed_plan.txt $P/${R}_bench_step_timeline_fixed_plan.txt
[ -f $O/timeline_side_graph.txt ] && cp $O/timeline_side_graph.txt $P/${R}_bench_step_timeline_side_graph_under_profiler.txt
{
  echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, with --kernel-trace only) of: python bench.py --no-cpu-baseline --no-graph --steps 6 --warmup 2"
  echo "# values are KB per dispatch as reported; FETCH_SIZE must be DOUBLED on gfx950 (MI355X_MICROARCH.md, HBM section) -- calibrated in this repo on"
  echo "# bn_relu_max / bn_bwd_reduce-type streaming kernels whose traffic is known exactly (reported 131 MB vs 268 MB streamed)."
  echo "# kernel names: stream_kernel<AMODE, EPI, K/16, CK, WN, ASM> (row-streaming GEMM, mlp_stream.hip); gemm_kernel<AMODE, EPI, VEC, WGM, WGN, WM, WN, DEPTH, BF3, WS, TL>; dw_ws_kernel<XMODE, DYMODE, NTO, NTI>; dw_kernel<...> = f32-MFMA dW (gather layers)"
  python tools/pmc_summary.py $O/pmc_FETCH_SIZE 24
  python tools/pmc_summary.py $O/pmc_WRITE_SIZE 24
} > $P/${R}_bench_pmc_fetch_write.txt
{
  echo "# rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (one pass, --kernel-trace only) of:"
  echo "#   python bench.py --no-cpu-baseline --no-graph --steps 6 --warmup 2     (sums over all dispatches of the run)"
  echo "# matrix-pipe utilisation of a kernel = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs): the counter advances 32 per"
  echo "# v_mfma_f32_32x32x16_bf16 and 64 per v_mfma_f32_32x32x2_f32 (MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over the 8 XCDs."
  python tools/pmc_table.py $O/pmc_sq 30
} > $P/${R}_bench_pmc_sq.txt
python tools/pmc_family.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $P/${R}_bench_pmc_family.json
[ -d $O/pmc_lds ] && { echo "# rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_WAVES (own pass) of: python bench.py --no-cpu-baseline --no-graph --no-padded-leg --steps 6 --warmup 2"; python tools/pmc_table.py $O/pmc_lds 40; } > $P/${R}_bench_pmc_lds.txt
for c in msg_seg pfn basic; do
  cp $O/prof_stats_$c/run_kernel_stats.csv $P/${R}_cfg_${c}_kernel_stats.csv
  grep "^{" $O/bench_line_$c.json > $P/${R}_cfg_${c}_bench_line.json
done
[ -f $O/prof_stats_msg_seg_serial/run_kernel_stats.csv ] && cp $O/prof_stats_msg_seg_serial/run_kernel_stats.csv $P/${R}_cfg_msg_seg_serial_kernel_stats.csv
{
  echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of: python bench.py --config pfn --no-cpu-baseline --steps 6 --warmup 2  (KB per dispatch; FETCH_SIZE x2 on gfx950)"
  python tools/pmc_summary.py $O/pmc_pfn_FETCH_SIZE 8
  python tools/pmc_summary.py $O/pmc_pfn_WRITE_SIZE 8
} > $P/${R}_cfg_pfn_pmc.txt
for c in pfn msg_seg basic; do
  python tools/pmc_family.py $O/pmc_${c}_FETCH_SIZE $O/pmc_${c}_WRITE_SIZE > $P/${R}_cfg_${c}_pmc_family.json
done
wc -l $P/${R}_*
