# round 4, step 4: the fused pairs on 16 x 16 x 32 too; A/Bs on one box: EV_H16Q=0, the hardware-cosine SnakeBeta build, a shorter hand-off spin limit
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s4; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py tests/test_gpu_precision.py tests/test_gpu_variants.py -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err; }
run q A=1
run noq EV_H16Q=0
run sin2hw EV_LIB_PATH=$PWD/emojivoice_amd/lib_ab/libev_sin2hw.so
run spin300 EV_SK_SPIN=300
run q2 A=1
python - <<'PY'
import json
for f in ("q", "noq", "sin2hw", "spin300", "q2"):
    try:
        d = json.loads(open(f"gpurun_out/s4/bench_{f}.json").read().strip().split("\n")[-1])
        print(f, d["value"], d["ms_per_step"], d.get("serial_ms_per_step"), d.get("stage_ms"))
    except Exception as e:
        print(f, "failed", e)
PY
timeout -k 10 300 python tools/shape_profile.py 64 $O/shape_q.txt > $O/shape_q.log 2>&1
EV_LIB_PATH=$PWD/emojivoice_amd/lib_ab/libev_sin2hw.so timeout -k 10 300 python tools/shape_profile.py 64 $O/shape_sin2hw.txt > $O/shape_sin2hw.log 2>&1
grep -h "pair\|lnff" $O/shape_q.txt $O/shape_sin2hw.txt
