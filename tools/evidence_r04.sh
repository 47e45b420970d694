# Round-4 evidence run (one gpurun call): kernel trace + stats of the bench, the PMC passes (busy, FETCH_SIZE, WRITE_SIZE, each on its own),
# per-shape HIP-event profile, the default bench line, configs 4 and 5, a soak with the hand-off counters.  MODE=lean skips the long tail.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ev4; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_b64.log 2>&1 &&
python tools/trace_steps.py $O/stats 416 7 > $O/conv_per_pass.txt 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/busy -- python3 bench.py --plain --no-pipeline --steps 1 --warmup 1 > $O/busy.log 2>&1 &&
python tools/pmc_summary.py $O/busy > $O/mfma_busy_pmc.txt 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch -- python3 bench.py --plain --no-pipeline --steps 1 --warmup 1 > $O/fetch.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write -- python3 bench.py --plain --no-pipeline --steps 1 --warmup 1 > $O/write.log 2>&1 &&
python tools/traffic_summary.py $O/fetch $O/write 2 > $O/traffic.json 2>&1
find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*counter_collection.csv" -size +2M -delete
python tools/shape_profile.py 64 $O/shape.txt > $O/shape.log 2>&1
EV_SPLIT=0 python tools/shape_profile.py 64 $O/shape_fp32.txt > $O/shape_fp32.log 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
if [ "$MODE" != "lean" ]; then
  python bench.py --config 4 --steps 3 > $O/config4.json 2> $O/config4.err
  python bench.py --config 5 > $O/config5.json 2> $O/config5.err
  python tools/latency_split.py > $O/batch1_latency_split.txt 2>&1
  timeout -k 10 600 python tools/fuzz_h16.py 80 1 > $O/fuzz_h16.txt 2>&1
  python bench.py --steps 200 --warmup 5 --no-extras --no-cpu-baseline > $O/soak200.json 2> $O/soak200.err
fi
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
ls -la $O | head -40
