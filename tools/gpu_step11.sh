# round 4, step 11: work stealing of the balanced launches + start stagger A/B
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s11; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_variants.py tests/test_gpu_configs.py -q -m gpu -x -k "balanced or b64 or config2 or ln_mlp or two_rank or arithmetic" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err; }
run steal A=1
run nosteal EV_NO_SK_STEAL=1
run stag2 EV_BAL_STAGGER=2
run stag4 EV_BAL_STAGGER=4
run steal2 A=1
python - <<'PY'
import json
for f in ("steal", "nosteal", "stag2", "stag4", "steal2"):
    try:
        d = json.loads(open(f"gpurun_out/s11/bench_{f}.json").read().strip().split("\n")[-1])
        print(f, d["value"], d["ms_per_step"], d.get("serial_ms_per_step"), d.get("stage_ms"), {k: v for k, v in d.get("balanced_handoffs", {}).items() if k != "note"})
    except Exception as e:
        print(f, "failed", e)
PY
