# Rehearsal of the N>1 training loops on a ONE-GPU box: two ranks on cuda:0 over gloo (MMVQA_REHEARSE_GLOO=1).
# Fails (non-zero exit) when a mode crashes, times out or leaves no checkpoint behind.
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
MINI="--resnet_layers 1 1 1 1 --resnet_width 8 --hidden_size 96 --n_layers 2 --vocab_size 64 --emb_vocab 64 --image_size 32 --steps_per_epoch 4 --val_steps 2 --epochs 2 --max_position_embeddings 16 --hidden_dropout_prob 0.1"
rc=0
for mode in supcon mlm vqa; do
  extra=""
  [ $mode = supcon ] && extra="--transformer_model realformer --batch_size 8"
  [ $mode = vqa ] && extra="--loss ASLSingleLabel --num_classes 11 --batch_size 4"
  [ $mode = mlm ] && extra="--batch_size 4"
  rm -rf /tmp/ddp_$mode
  log=/tmp/ddp_$mode.log
  if ! MMVQA_REHEARSE_GLOO=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) -m mmvqa_amd.train $mode --lr 1e-3 --save_dir /tmp/ddp_$mode $extra $MINI > $log 2>&1; then
    echo "FAILED: $mode (exit status of torch.distributed.run)"; tail -20 $log; rc=1; continue
  fi
  grep -E "Epoch|data parallel" $log | head -6 || true
  if grep -qE "Traceback|Error" $log; then echo "FAILED: $mode (error text in the log)"; rc=1; fi
  want=/tmp/ddp_$mode/MLM/run.pt
  [ -s $want ] || { echo "FAILED: $mode left no checkpoint at $want"; rc=1; }
done
exit $rc
