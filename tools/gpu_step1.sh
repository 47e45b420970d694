# round 4, step 1: full GPU suite on the spill-free library, bench, and the -fno-slp-vectorize A/B (same box)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s1; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err
EV_LIB_PATH=$PWD/emojivoice_amd/lib_ab/libev_noslp.so timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > $O/bench_noslp.json 2> $O/bench_noslp.err
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > $O/bench2.json 2> $O/bench2.err
python - <<'PY'
import json
for f in ("bench", "bench_noslp", "bench2"):
    try:
        d = json.loads(open(f"gpurun_out/s1/{f}.json").read().strip().split("\n")[-1])
        print(f, d["value"], d["ms_per_step"], d.get("serial_ms_per_step"), d.get("stage_ms"))
    except Exception as e:
        print(f, "failed", e)
PY
