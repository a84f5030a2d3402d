set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
tail -c 600 gpurun_out/r3a/bench.json
rocprofv3 --kernel-trace --stats -d gpurun_out/r3a/prof -o t --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r3a/prof.log 2>&1
find gpurun_out/r3a/prof -name "*kernel_trace.csv" | head -1 > gpurun_out/r3a/trace_path.txt
python tools/trace_timeline.py $(cat gpurun_out/r3a/trace_path.txt) gpurun_out/r3a/timeline.json 2
python tools/trace_steady.py $(cat gpurun_out/r3a/trace_path.txt) gpurun_out/r3a/steady.csv 2
