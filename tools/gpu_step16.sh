set -x
O=gpurun_out/s16; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_precision.py -m gpu -x -q -k "attn or attention or ln_mlp" > $O/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -5 $O/pytest_ops.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "parity rc=$?"; tail -5 $O/pytest_parity.log
python bench.py --no-extras --no-cpu-baseline > $O/bench_h16attn.json 2> $O/bench_h16attn.err
EV_NO_ATTN_H16=1 python bench.py --no-extras --no-cpu-baseline > $O/bench_f32attn.json 2> $O/bench_f32attn.err
python bench.py --no-extras --no-cpu-baseline > $O/bench_h16attn2.json 2> $O/bench_h16attn2.err
EV_SP_NOVOC=1 EV_ATTN_STAMPS=$O/attn_stamps.txt python tools/shape_profile.py 64 $O/shape.txt > $O/shape.log 2>&1
python - <<'PY'
import json
for n in ("h16attn","f32attn","h16attn2"):
    try:
        d=json.loads(open(f"gpurun_out/s16/bench_{n}.json").read().strip().split("\n")[-1]); print(n, d["value"], d["ms_per_step"], d["serial_ms_per_step"], d["stage_ms"])
    except Exception as e: print(n, "failed", e)
PY
head -13 $O/attn_stamps.txt
