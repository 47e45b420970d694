set -x
O=gpurun_out/s19; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_precision.py -m gpu -x -q -k "chains" > $O/pytest_a.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_a.log
EV_CHAIN_MAXHALO=36 timeout -k 10 900 python -m pytest tests/test_gpu_precision.py tests/test_gpu_parity.py -m gpu -x -q -k "chains or vocoder or hifigan" > $O/pytest_b.log 2>&1; echo "pytest36 rc=$?"; tail -5 $O/pytest_b.log
python bench.py --no-extras --no-cpu-baseline > $O/bench_h12.json 2> $O/bench_h12.err
EV_CHAIN_MAXHALO=36 python bench.py --no-extras --no-cpu-baseline > $O/bench_h36.json 2> $O/bench_h36.err
python bench.py --no-extras --no-cpu-baseline > $O/bench_h12b.json 2> $O/bench_h12b.err
EV_CHAIN_MAXHALO=36 python tools/shape_profile.py 64 $O/shape_h36.txt > $O/shape_h36.log 2>&1
python - <<'PY'
import json
for n in ("h12","h36","h12b"):
    try:
        d=json.loads(open(f"gpurun_out/s19/bench_{n}.json").read().strip().split("\n")[-1]); print(n, d["value"], d["ms_per_step"], d["serial_ms_per_step"], d["stage_ms"])
    except Exception as e: print(n, "failed", e)
PY
grep "^pair" $O/shape_h36.txt
