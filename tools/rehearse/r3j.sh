set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_STAGE:-${GRAFT_REPO_ROOT:?}}"
mkdir -p gpurun_out/r3j
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -q -m gpu -k "fused_qkv or attention" > gpurun_out/r3j/ops.log 2>&1; tail -3 gpurun_out/r3j/ops.log
for d in 0 1 2 3; do MMVQA_QA_DBG=$d timeout -k 10 100 python tools/qkvattn_bench.py 2>&1 | tail -1; done
bash tools/rehearse/r3_evidence_b.sh
