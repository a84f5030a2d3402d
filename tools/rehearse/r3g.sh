set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_STAGE:-${GRAFT_REPO_ROOT:?}}"
mkdir -p gpurun_out/r3g
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -q -m gpu -k "fused_qkv or attention" > gpurun_out/r3g/ops.log 2>&1; tail -8 gpurun_out/r3g/ops.log
timeout -k 10 600 python -m pytest tests/test_hip_model.py tests/test_hip_loops.py -q -m gpu -x > gpurun_out/r3g/model.log 2>&1; tail -5 gpurun_out/r3g/model.log
for rep in 1 2; do
for v in "MMVQA_PERSIST_KINDS=3" "MMVQA_NO_PERSIST=1" "MMVQA_PERSIST_KINDS=1" "MMVQA_PERSIST_KINDS=2" "MMVQA_NO_FUSED_QKV=1 MMVQA_NO_PERSIST=1"; do
  tag=$(echo "$v" | tr ' =' '__')_$rep
  env $v timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r3g/bench_$tag.json 2> gpurun_out/r3g/bench_$tag.err || echo "bench $tag failed"
  python -c "
import json
d=json.load(open('gpurun_out/r3g/bench_$tag.json')); print('$tag', round(d['ms_per_step'],3), round(d['value'],1))" || tail -5 gpurun_out/r3g/bench_$tag.err
done
done
