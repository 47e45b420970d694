# round 4, step 3: the 16 x 16 x 32 form of conv_h16_kernel: correctness (ops + parity + precision tests), A/B against the 32 x 32 x 16 form on one box
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s3; rm -rf $O; mkdir -p $O
timeout -k 10 60 tools/mfma_shape_probe 2>&1 | head -3 > $O/layout.txt; cat $O/layout.txt
timeout -k 10 60 tools/sin2_probe > $O/sin2_probe.txt 2>&1; cat $O/sin2_probe.txt
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py tests/test_gpu_precision.py tests/test_gpu_variants.py -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > $O/bench_q.json 2> $O/bench_q.err
EV_H16Q=0 timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > $O/bench_noq.json 2> $O/bench_noq.err
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > $O/bench_q2.json 2> $O/bench_q2.err
python - <<'PY'
import json
for f in ("bench_q", "bench_noq", "bench_q2"):
    try:
        d = json.loads(open(f"gpurun_out/s3/{f}.json").read().strip().split("\n")[-1])
        print(f, d["value"], d["ms_per_step"], d.get("serial_ms_per_step"), d.get("stage_ms"), d.get("value_fp32_mfma"))
    except Exception as e:
        print(f, "failed", e)
PY
timeout -k 10 300 python tools/shape_profile.py 64 $O/shape_q.txt > $O/shape_q.log 2>&1
EV_H16Q=0 timeout -k 10 300 python tools/shape_profile.py 64 $O/shape_noq.txt > $O/shape_noq.log 2>&1
