set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_STAGE:-${GRAFT_REPO_ROOT:?}}"
mkdir -p gpurun_out/r3l
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -q -m gpu -k "fused_qkv or attention" > gpurun_out/r3l/ops.log 2>&1; tail -3 gpurun_out/r3l/ops.log
for d in 0 1; do MMVQA_QA_DBG=$d timeout -k 10 100 python tools/qkvattn_bench.py 2>&1 | tail -1; done
bash tools/rehearse/r3_evidence_a.sh
timeout -k 10 900 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r3ev/round3_gpu_suite.txt 2>&1; tail -3 gpurun_out/r3ev/round3_gpu_suite.txt
