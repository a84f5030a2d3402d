#!/bin/bash
# per-shape ranking of the GEMM launches of one config-2 step (tools/igemm_rank.py); run on the GPU box
set -eo pipefail
cd "${GRAFT_STAGE:-${GRAFT_REPO_ROOT:?}}"
out=gpurun_out/${1:-rank}; mkdir -p "$out"
export TMPDIR=/tmp
MMVQA_IGEMM_LOG=1 rocprofv3 --kernel-trace -d "$out/p" -o t --output-format csv -- python3 bench.py ${RANK_ARGS:-} --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2> "$out/log.txt" > "$out/bench.out"
python tools/igemm_rank.py "$(find "$out/p" -name 't_kernel_trace.csv' | head -1)" "$out/log.txt" 70 > "$out/rank.txt"
rm -rf "$out/p"
tail -5 "$out/rank.txt"
