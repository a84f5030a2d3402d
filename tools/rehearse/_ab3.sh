cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_STAGE"
run() { (cd $1 && shift && env "$@" python bench.py --config ${CFG:-3} --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"); }
for rep in 1 2; do
  echo "cfg${CFG:-3} fold: $(run . MMVQA_X=1)"
  echo "cfg${CFG:-3} no fold: $(run . MMVQA_NO_BN_FOLD=1)"
done
