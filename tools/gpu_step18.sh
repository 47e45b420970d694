set -x
O=gpurun_out/s18; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_precision.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_a.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_a.log
python bench.py --no-extras --no-cpu-baseline > $O/bench_chain.json 2> $O/bench_chain.err
EV_NO_CHAIN=1 python bench.py --no-extras --no-cpu-baseline > $O/bench_nochain.json 2> $O/bench_nochain.err
python bench.py --no-extras --no-cpu-baseline > $O/bench_chain2.json 2> $O/bench_chain2.err
python tools/shape_profile.py 64 $O/shape.txt > $O/shape.log 2>&1
EV_NO_CHAIN=1 python tools/shape_profile.py 64 $O/shape_nochain.txt > $O/shape_nochain.log 2>&1
python - <<'PY'
import json
for n in ("chain","nochain","chain2"):
    try:
        d=json.loads(open(f"gpurun_out/s18/bench_{n}.json").read().strip().split("\n")[-1]); print(n, d["value"], d["ms_per_step"], d["serial_ms_per_step"], d["stage_ms"], d["cpu_baseline"] if "cpu_baseline" in d and d["cpu_baseline"] else "")
    except Exception as e: print(n, "failed", e)
PY
grep "^pair" $O/shape.txt; echo ---; grep "^pair" $O/shape_nochain.txt
