cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_STAGE"
run() { env "$@" python bench.py --config ${CFG:-2} --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"; }
for rep in 1 2 3 4; do
  echo "cfg${CFG:-2} default(0.92): $(run MMVQA_X=1)"
  echo "cfg${CFG:-2} gain 0.85: $(run MMVQA_SK_GAIN=0.85)"
  echo "cfg${CFG:-2} gain 0.75: $(run MMVQA_SK_GAIN=0.75)"
  echo "cfg${CFG:-2} finish form: $(run MMVQA_SK_FINISH=1)"
done
