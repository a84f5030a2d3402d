set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_STAGE:-${GRAFT_REPO_ROOT:?}}"
mkdir -p gpurun_out/r3h
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -q -m gpu -k "fused_qkv or attention" > gpurun_out/r3h/ops.log 2>&1; tail -5 gpurun_out/r3h/ops.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3h/bench.json 2> gpurun_out/r3h/bench.err || tail -5 gpurun_out/r3h/bench.err
python -c "
import json
d=json.load(open('gpurun_out/r3h/bench.json')); print(round(d['ms_per_step'],3), round(d['value'],1)); b=d['roofline']['blocks']
print({k:(round(v.get('frac_of_f32_mfma_peak',0),3), round(v.get('ms',0),4)) for k,v in b.items() if k.startswith(('qkv','att'))})"
