#!/bin/bash
# Copies the summaries of tools/evidence_r03.sh (gpurun_out/ev3) into profiles/ under their committed names.
set -e
O=gpurun_out/ev3; P=profiles
cp $O/bench_default.json $P/r03_bench_default.json
cp $O/bench_bf16x6.json $P/r03_bench_EV_SPLIT6_same_box.json
cp $O/bench_fp32_mfma.json $P/r03_bench_EV_SPLIT0_same_box.json
cp $O/bench_b64.log $P/r03_bench_b64.log
cp $O/conv_per_pass.txt $P/r03_bench_b64_conv_per_pass.txt
cp "$(ls -t $O/stats/*/*kernel_stats.csv | head -1)" $P/r03_bench_b64_kernel_stats.csv
cp $O/mfma_busy_pmc.txt $P/r03_mfma_busy_pmc.txt
cp $O/traffic.json $P/r03_conv_hbm_traffic_pmc.json
cp $O/shape.txt $P/r03_conv_per_shape_hip_events.txt
cp $O/shape_fp32.txt $P/r03_conv_per_shape_EV_SPLIT0_same_box.txt
cp $O/shape_bf16x6.txt $P/r03_conv_per_shape_EV_SPLIT6_same_box.txt
grep -v "amdgpu.ids" $O/arith_accuracy.txt > $P/r03_arith_accuracy_vs_fp64.txt
grep -v "amdgpu.ids" $O/conv_split_ablation.txt > $P/r03_conv_arithmetic_settings_per_layer.txt
cp $O/config4.json $P/r03_config4_ode_sweep.json
cp $O/config5.json $P/r03_config5_streaming.json
grep -v "amdgpu.ids" $O/batch1_latency_split.txt > $P/r03_batch1_latency_split.txt
grep -v "amdgpu.ids" $O/fuzz_h16.txt > $P/r03_fuzz_h16_vs_fp32_mfma.txt
