cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_STAGE"
run() { python bench.py --config ${CFG:-2} --steps 20 --warmup 3 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"; }
for rep in 1 2; do
  echo "cfg${CFG:-2} adam after backward: $(run --no-overlap-adam)"
  for w in 64 256 1024; do echo "cfg${CFG:-2} beside, $w WGs: $(MMVQA_ADAM_WGS=$w run)"; done
done
