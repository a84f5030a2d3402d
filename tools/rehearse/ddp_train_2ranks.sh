set -e
cd $GRAFT_REPO_ROOT
MINI="--resnet_layers 1 1 1 1 --resnet_width 8 --hidden_size 96 --n_layers 2 --vocab_size 64 --emb_vocab 64 --image_size 32 --steps_per_epoch 4 --val_steps 2 --epochs 2 --max_position_embeddings 16 --hidden_dropout_prob 0.1"
for mode in supcon mlm vqa; do
  extra=""
  [ $mode = supcon ] && extra="--transformer_model realformer --batch_size 8"
  [ $mode = vqa ] && extra="--loss ASLSingleLabel --num_classes 11 --batch_size 4"
  [ $mode = mlm ] && extra="--batch_size 4"
  MMVQA_REHEARSE_GLOO=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) -m mmvqa_amd.train $mode --lr 1e-3 --save_dir /tmp/ddp_$mode $extra $MINI 2>&1 | grep -E "Epoch|Error|error|Traceback" | head -6
done
ls /tmp/ddp_supcon /tmp/ddp_supcon/MLM
