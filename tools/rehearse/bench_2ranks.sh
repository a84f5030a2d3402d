# Rehearsal of bench.py's N>1 path on a ONE-GPU box: two ranks on cuda:0 over gloo (MMVQA_REHEARSE_GLOO=1), configs 2-5,
# default and with --overlap-adam (Adam behind each bucket's all-reduce).  Not a measurement.
set -o pipefail
cd "${GRAFT_STAGE:-${GRAFT_REPO_ROOT:?}}"
rc=0
for cfg in 2 3 4 5; do
  for extra in "" "--overlap-adam"; do
    log=/tmp/b2_${cfg}_${extra#--}.log
    if ! MMVQA_REHEARSE_GLOO=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29300 + RANDOM % 300)) bench.py --gpus 2 --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-roofline $extra > $log 2>&1; then
      echo "FAILED cfg $cfg $extra"; tail -15 $log; rc=1; continue
    fi
    python - "$log" "$cfg" "$extra" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
d = json.loads(line)
print("cfg", sys.argv[2], sys.argv[3] or "(default)", "n_gpus", d["n_gpus"], round(d["ms_per_step"], 1), "ms", d["config"].get("comm", {}).get("backend"), d["config"].get("comm", {}).get("buckets"))
PY
  done
done
exit $rc
