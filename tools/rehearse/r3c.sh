set -e -o pipefail
cd "${GRAFT_STAGE:-${GRAFT_REPO_ROOT:?}}"
mkdir -p gpurun_out/r3c
timeout -k 10 900 python -m pytest tests -q -m gpu --durations=15 > gpurun_out/r3c/tests.log 2>&1 || { tail -40 gpurun_out/r3c/tests.log; exit 1; }
tail -25 gpurun_out/r3c/tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3c/bench.json 2> gpurun_out/r3c/bench.err
MMVQA_NO_BN_FOLD=1 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r3c/bench_nofold.json 2> gpurun_out/r3c/bench_nofold.err
MMVQA_BN_SLOTS=16 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r3c/bench_slots16.json 2> gpurun_out/r3c/bench_slots16.err
python - <<'PY'
import json
for n in ("bench", "bench_nofold", "bench_slots16"):
    d = json.load(open(f"gpurun_out/r3c/{n}.json"))
    print(n, round(d["ms_per_step"], 3), round(d["value"], 1))
PY
