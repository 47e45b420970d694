set -x
O=gpurun_out/s17; rm -rf $O; mkdir -p $O
python bench.py --no-extras --no-cpu-baseline > $O/bench_pad.json 2> $O/bench_pad.err
EV_RN0_PAD=0 python bench.py --no-extras --no-cpu-baseline > $O/bench_nopad.json 2> $O/bench_nopad.err
python bench.py --no-extras --no-cpu-baseline > $O/bench_pad2.json 2> $O/bench_pad2.err
EV_SP_NOVOC=1 python tools/shape_profile.py 64 $O/shape.txt > $O/shape.log 2>&1
python - <<'PY'
import json
for n in ("pad","nopad","pad2"):
    try:
        d=json.loads(open(f"gpurun_out/s17/bench_{n}.json").read().strip().split("\n")[-1]); print(n, d["value"], d["ms_per_step"], d["serial_ms_per_step"], d["stage_ms"])
    except Exception as e: print(n, "failed", e)
PY
cat $O/shape.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
