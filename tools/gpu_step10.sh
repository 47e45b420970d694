# round 4, step 10: full GPU suite + default bench line with everything of the round (amax slots v3, decode graphs with kernel-only captures)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s10; rm -rf $O; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
timeout -k 10 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/s10/bench_default.json").read().strip().split("\n")[-1])
print("default", d["value"], d["ms_per_step"], d["stage_ms"], "fp32:", d.get("value_fp32_mfma", {}).get("value"), d.get("balanced_handoffs"))
print("config5", json.dumps(d.get("config5")))
PY
EV_SK_SPIN=300 timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > $O/bench_spin300.json 2> $O/bench_spin300.err
EV_NO_SK_BALANCE=1 timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > $O/bench_nobal.json 2> $O/bench_nobal.err
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > $O/bench_ref.json 2> $O/bench_ref.err
python - <<'PY'
import json
for f in ("spin300", "nobal", "ref"):
    try:
        d = json.loads(open(f"gpurun_out/s10/bench_{f}.json").read().strip().split("\n")[-1])
        print(f, d["value"], d["ms_per_step"], d.get("stage_ms"), d.get("balanced_handoffs"))
    except Exception as e:
        print(f, "failed", e)
PY
