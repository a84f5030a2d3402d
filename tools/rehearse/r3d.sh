set -o pipefail
cd "${GRAFT_STAGE:-${GRAFT_REPO_ROOT:?}}"
mkdir -p gpurun_out/r3d
timeout -k 10 600 python -m pytest tests/test_hip_ops.py -q -m gpu --durations=5 -k "persistent or folded or conv_ or linear" > gpurun_out/r3d/ops.log 2>&1
tail -15 gpurun_out/r3d/ops.log
timeout -k 10 900 python -m pytest tests -q -m gpu --durations=10 --deselect tests/test_hip_ops.py > gpurun_out/r3d/tests.log 2>&1
tail -15 gpurun_out/r3d/tests.log
for v in "" "MMVQA_NO_PERSIST=1" "MMVQA_NO_BN_FOLD=1 MMVQA_NO_PERSIST=1" "MMVQA_BN_SLOTS=16 MMVQA_NO_PERSIST=1"; do
  tag=$(echo "base $v" | tr ' =' '__')
  env $v timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r3d/bench_$tag.json 2> gpurun_out/r3d/bench_$tag.err || echo "bench $tag failed"
  python -c "
import json,sys
d=json.load(open('gpurun_out/r3d/bench_$tag.json')); print('$tag', round(d['ms_per_step'],3), round(d['value'],1))" || tail -5 gpurun_out/r3d/bench_$tag.err
done
