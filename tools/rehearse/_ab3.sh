cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_STAGE"
run() { env "$@" python bench.py --config ${CFG:-2} --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"; }
for rep in 1 2 3 4 5; do
  echo "cfg${CFG:-2} encoder wgrads beside: $(run MMVQA_X=1)"
  echo "cfg${CFG:-2} encoder one stream: $(run MMVQA_ENC_SIDE_OFF=1)"
done
