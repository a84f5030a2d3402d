set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_STAGE:-${GRAFT_REPO_ROOT:?}}"
mkdir -p gpurun_out/r3f
timeout -k 10 900 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r3f/tests.log 2>&1; tail -6 gpurun_out/r3f/tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3f/bench.json 2> gpurun_out/r3f/bench.err || tail -5 gpurun_out/r3f/bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r3f/prof -o t --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r3f/prof.log 2>&1
tr=$(find gpurun_out/r3f/prof -name "*kernel_trace.csv" | head -1)
python tools/trace_timeline.py $tr gpurun_out/r3f/timeline.json 2
python tools/trace_steady.py $tr gpurun_out/r3f/steady.csv 2
timeout -k 10 200 python tools/jpeg_decode_rate.py --images 128 > gpurun_out/r3f/jpeg.json 2> gpurun_out/r3f/jpeg.err || tail -3 gpurun_out/r3f/jpeg.err
python -c "
import json
d=json.load(open('gpurun_out/r3f/bench.json')); print(round(d['ms_per_step'],3), round(d['value'],1), round(d['roofline']['frac'],4), round(d['roofline']['single_stream']['frac'],4))
b=d['roofline']['blocks']
print({k:(round(v.get('frac_of_f32_mfma_peak',0),3), round(v.get('ms',0),3)) for k,v in b.items()})"
