# Same-box A/B of bench.py between the working tree and older builds unpacked under tools/build/<name>/ (whole trees with
# their own built library): alternating runs.   REPS=2 bash tools/rehearse/ab_trees.sh . tools/build/prev_a tools/build/prev_b
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_STAGE:-${GRAFT_REPO_ROOT:?}}"
R=$PWD; O=$R/gpurun_out/abt; mkdir -p $O
for rep in $(seq 1 ${REPS:-2}); do
  for t in "$@"; do
    tag=$(echo "$t" | tr '/.' '__')_$rep
    d=$R/$t; [ -d "$d" ] || d=$GRAFT_REPO_ROOT/$t   # (tools/build/ is not part of a staged copy)
    (cd $d && timeout -k 10 300 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-roofline ${BENCH_ARGS:-} > $O/bench_$tag.json 2> $O/bench_$tag.err) || echo "bench $tag failed"
    python -c "
import json
d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print('$t', $rep, round(d['ms_per_step'],3), 'ms/step')" || tail -5 $O/bench_$tag.err
  done
done
