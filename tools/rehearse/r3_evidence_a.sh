# Round-3 evidence, part A (config 2): the driver's bench line, kernel trace -> timeline + steady statistics, PMC passes.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_STAGE:-${GRAFT_REPO_ROOT:?}}"
O=gpurun_out/r3ev; mkdir -p $O
timeout -k 10 500 python bench.py > $O/round3_bench_cfg2.json 2> $O/bench_cfg2.err || tail -5 $O/bench_cfg2.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof2 -o cfg2 --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/prof2.log 2>&1
tr=$(find $O/prof2 -name "*kernel_trace.csv" | head -1)
python tools/trace_timeline.py $tr $O/round3_timeline_cfg2.json 2
python tools/trace_steady.py $tr $O/round3_kernel_stats_steady_cfg2.csv 2
cp $(find $O/prof2 -name "*kernel_stats.csv" | head -1) $O/round3_kernel_stats_bench_cfg2.csv
rm -rf $O/prof2
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/f2 --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/f2.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/w2 --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/w2.log 2>&1
python tools/pmc_traffic.py $O/f2 $O/w2 $O/round3_igemm_traffic.json > /dev/null && rm -rf $O/f2 $O/w2
MMVQA_IGEMM_LOG=1 timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES -d $O/m2 --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/m2.out 2> $O/m2.log
python tools/pmc_mfma_util.py $O/m2 $O/m2.log $O/round3_mfma_util.json > /dev/null && rm -rf $O/m2 $O/m2.log
ls -la $O
