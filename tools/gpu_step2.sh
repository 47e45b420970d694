# round 4, step 2: full GPU suite (incl. the new precision tests), dynamic-range table, stamps of the spill-free kernels, MFMA shape probe
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s2; rm -rf $O; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout -k 10 300 python tools/dynamic_range.py > $O/dynamic_range.txt 2>&1
EV_MLP_STAMPS=$O/mlp_stamps.txt EV_BAL_STAMPS=$O/bal_stamps.txt timeout -k 10 300 python bench.py --plain --no-pipeline --steps 1 --warmup 0 > $O/stamps.log 2>&1
timeout -k 10 120 tools/mfma_shape_probe > $O/mfma_shape_probe.txt 2>&1
cat $O/mfma_shape_probe.txt
cat $O/dynamic_range.txt
