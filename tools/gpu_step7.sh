# round 4, step 7: amax slots v3 (pairs bound their residual by their own tile maximum, emit only into consumed sums), decode graphs: full suite, A/B, config 5
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s7; rm -rf $O; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err; }
run amax A=1
run noamax EV_NO_AMAX=1
run amax2 A=1
python - <<'PY'
import json
for f in ("amax", "noamax", "amax2"):
    try:
        d = json.loads(open(f"gpurun_out/s7/bench_{f}.json").read().strip().split("\n")[-1])
        print(f, d["value"], d["ms_per_step"], d.get("serial_ms_per_step"), d.get("stage_ms"))
    except Exception as e:
        print(f, "failed", e)
PY
timeout -k 10 300 python tools/shape_profile.py 64 $O/shape_amax.txt > $O/shape_amax.log 2>&1
EV_NO_AMAX=1 timeout -k 10 300 python tools/shape_profile.py 64 $O/shape_noamax.txt > $O/shape_noamax.log 2>&1
paste <(head -22 $O/shape_noamax.txt | cut -c1-75) <(head -22 $O/shape_amax.txt | awk '{print $(NF-1), $NF}')
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/s7/bench_default.json").read().strip().split("\n")[-1])
print("default", d["value"], d["ms_per_step"], d["stage_ms"], "fp32:", d.get("value_fp32_mfma", {}).get("value"))
print("config5", json.dumps(d.get("config5")))
print("cpu", d.get("cpu_baseline"))
PY
