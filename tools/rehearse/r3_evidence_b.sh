# Round-3 evidence, part B: configs 3, 4, 5 (builder-run), input pipeline, GPU suite summary.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_STAGE:-${GRAFT_REPO_ROOT:?}}"
O=gpurun_out/r3ev; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu --durations=8 > $O/round3_gpu_suite.txt 2>&1; tail -3 $O/round3_gpu_suite.txt
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/f3 --output-format csv -- python3 bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/f3.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/w3 --output-format csv -- python3 bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/w3.log 2>&1
python tools/pmc_traffic.py $O/f3 $O/w3 $O/round3_igemm_traffic_cfg3.json > /dev/null && rm -rf $O/f3 $O/w3
mkdir -p profiles_tmp && cp $O/round3_igemm_traffic_cfg3.json profiles/ 2>/dev/null
for c in 3 4 5; do
  timeout -k 10 400 python bench.py --config $c --no-cpu-baseline > $O/round3_bench_cfg$c.json 2> $O/bench_cfg$c.err || tail -5 $O/bench_cfg$c.err
done
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof3 -o cfg3 --output-format csv -- python3 bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/prof3.log 2>&1
tr=$(find $O/prof3 -name "*kernel_trace.csv" | head -1)
python tools/trace_steady.py $tr $O/round3_kernel_stats_steady_cfg3.csv 2
python tools/trace_timeline.py $tr $O/round3_timeline_cfg3.json 2
rm -rf $O/prof3
timeout -k 10 300 python tools/jpeg_decode_rate.py --images 128 > $O/round3_jpeg_decode.json 2> $O/jpeg.err || tail -3 $O/jpeg.err
timeout -k 10 300 python tools/augment_bench.py > $O/round3_augment_bench.json 2> $O/aug.err || tail -3 $O/aug.err
ls -la $O
