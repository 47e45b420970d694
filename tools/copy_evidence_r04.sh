#!/bin/bash
# Copies the summaries of tools/evidence_r04.sh (gpurun_out/ev4) into profiles/ under their committed names.
set -e
O=gpurun_out/ev4; P=profiles
cp $O/bench_default.json $P/r04_bench_default.json
cp $O/bench_b64.log $P/r04_bench_b64.log
cp $O/conv_per_pass.txt $P/r04_bench_b64_conv_per_pass.txt
cp "$(ls -t $O/stats/*/*kernel_stats.csv | head -1)" $P/r04_bench_b64_kernel_stats.csv
cp $O/mfma_busy_pmc.txt $P/r04_mfma_busy_pmc.txt
cp $O/traffic.json $P/r04_conv_hbm_traffic_pmc.json
cp $O/shape.txt $P/r04_conv_per_shape_hip_events.txt
cp $O/shape_fp32.txt $P/r04_conv_per_shape_EV_SPLIT0_same_box.txt
[ -f $O/config4.json ] && cp $O/config4.json $P/r04_config4_ode_sweep.json
[ -f $O/config5.json ] && cp $O/config5.json $P/r04_config5_streaming.json
[ -f $O/batch1_latency_split.txt ] && grep -v "amdgpu.ids" $O/batch1_latency_split.txt > $P/r04_batch1_latency_split.txt
[ -f $O/fuzz_h16.txt ] && grep -v "amdgpu.ids" $O/fuzz_h16.txt > $P/r04_fuzz_h16_vs_fp32_mfma.txt
[ -f $O/soak200.json ] && cp $O/soak200.json $P/r04_soak_200_steps.json
true
