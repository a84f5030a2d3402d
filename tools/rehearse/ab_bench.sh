# Same-box A/B of bench.py under environment switches: alternating runs, REPS repetitions.
#   REPS=4 bash tools/rehearse/ab_bench.sh "MMVQA_X=1" "MMVQA_NO_BN_FOLD=1" ["MMVQA_PERSIST_KINDS=3" ...]
# (MMVQA_X=1 is a no-op variable: the default build.)  Run it through gpurun; results also land in gpurun_out/ab/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_STAGE:-${GRAFT_REPO_ROOT:?}}"
O=gpurun_out/ab; mkdir -p $O
for rep in $(seq 1 ${REPS:-2}); do
  for v in "$@"; do
    tag=$(echo "$v" | tr ' =' '__')_$rep
    env $v timeout -k 10 300 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_$tag.json 2> $O/bench_$tag.err || echo "bench $tag failed"
    python -c "
import json
d=json.load(open('$O/bench_$tag.json')); print('$tag', round(d['ms_per_step'],3), 'ms/step', round(d['value'],1), 'samples/s')" || tail -5 $O/bench_$tag.err
  done
done
