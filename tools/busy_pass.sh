set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/busy; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/busy -- python3 bench.py --plain --no-pipeline --steps 1 --warmup 1 > $O/busy.log 2>&1
python tools/pmc_summary.py $O/busy > $O/mfma_busy_pmc.txt 2>&1
find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*counter_collection.csv" -size +2M -delete
