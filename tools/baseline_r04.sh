# round-4 baseline of the library as it stood at the start of the round: GPU tests, the default bench line, per-shape table, stamps
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/base4; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err && tail -c 1500 $O/bench.json
EV_MLP_STAMPS=$O/mlp_stamps.txt EV_BAL_STAMPS=$O/bal_stamps.txt EV_ATTN_STAMPS=$O/attn_stamps.txt EV_QKV_STAMPS=$O/qkv_stamps.txt timeout -k 10 300 python bench.py --plain --no-pipeline --steps 1 --warmup 0 > $O/stamps.log 2>&1
timeout -k 10 300 python tools/shape_profile.py 64 $O/shape.txt > $O/shape.log 2>&1
ls -la $O
