"""Training loops (callers of the hot path) -- build-owned counterparts of the reference's scripts,
which never ship to the GPU box (SURVEY.md 8(b) "Callers the build must supply"):

  mlm      pretrain/roco_train.py:155-197 + pretrain/roco_utils.py:207-372 (train_one_epoch / validate)
  supcon   pretrain/roco_supcon_train.py:137,168-202 + models/SupConLoss/supcon_utils.py:253-379
  vqa      vqamed2019/train.py:125-296 + vqamed2019/utils.py:625-767
  eval     vqamed2019/eval.py:99-180 + vqamed2019/utils.py:769-843 (test-set run: metrics, <model>_preds.csv, <model>_res.txt)

Kept from the reference: option names and defaults, Adam(lr) + ReduceLROnPlateau(patience, factor) on the
validation loss, zero_grad -> forward -> loss -> backward -> step order, loss / accuracy definitions,
half-batch x 2 views for SupCon, best-val-loss checkpoint, the 5-epoch "recorder" dict
{epoch, optimizer, scheduler, scaler, model}, --resume, the VQA early-stop counter and classifier[2] surgery.
Not kept (out of scope): real datasets/tokenizer/augmentation (synthetic batches with the same
layout stand in: mmvqa_amd.synth), wandb, BLEU.  One process per GPU under torch.distributed (RCCL).

    python -m mmvqa_amd.train mlm    --run_name r --mlm_prob 0.15 --epochs 2 --steps_per_epoch 20
    python -m mmvqa_amd.train supcon --run_name r --mlm_prob 0.15 --batch_size 32
    python -m mmvqa_amd.train vqa    --run_name r --loss ASLSingleLabel --batch_size 64
    python -m mmvqa_amd.train eval   --model_dir save/MLM/r.pt --num_classes 1552 --batch_size 16
"""
from __future__ import annotations

import argparse
import os
import sys

import torch
import torch.distributed as dist
from torch.optim import lr_scheduler

from . import FusedAdam, Model, asl_loss, checkpoint, evaluate, mlm_loss, split_feat, supcon_loss, synth
from .ddp import GradReducer, comm_info, global_supcon_views, sync_replicas


def common_args(p):
    p.add_argument("-r", "--run_name", type=str, default="run")
    p.add_argument("--save_dir", type=str, default="save")
    p.add_argument("--batch_size", type=int, default=16)
    p.add_argument("--patience", type=int, default=5)
    p.add_argument("--factor", type=float, default=0.1)
    p.add_argument("--epochs", type=int, default=10)
    p.add_argument("--steps_per_epoch", type=int, default=50, help="synthetic batches per epoch")
    p.add_argument("--val_steps", type=int, default=5)
    p.add_argument("--n_layers", type=int, default=4)
    p.add_argument("--heads", type=int, default=12)
    p.add_argument("--type_vocab_size", type=int, default=2)
    p.add_argument("--vocab_size", type=int, default=30522)
    p.add_argument("--hidden_size", type=int, default=768)
    p.add_argument("--hidden_dropout_prob", type=float, default=0.3)
    p.add_argument("--cnn_encoder", type=str, default="resnet152")
    p.add_argument("--transformer_model", type=str, default="transformer",
                   choices=["transformer", "realformer", "feedback-transformer"])
    p.add_argument("--num_vis", type=int, default=5)
    p.add_argument("--use_relu", action="store_true", default=False)
    p.add_argument("--resume", action="store_true", default=False)
    p.add_argument("--image_size", type=int, default=224)
    p.add_argument("--seed", type=int, default=1234)
    p.add_argument("--state_dict", type=str, default=None, help="local checkpoint to start from")
    p.add_argument("--backbone_weights", type=str, default=None,
                   help="local torchvision resnet152 / timm tf_efficientnetv2_m state_dict (what pretrained=True fetches, image_encoding.py:20-26)")
    p.add_argument("--bert_weights", type=str, default=None,
                   help="local HF bert-base-uncased state_dict: its embeddings are used (mmbert.py:52-56)")
    # reduced backbones for smoke tests
    p.add_argument("--resnet_layers", type=int, nargs=4, default=[3, 8, 36, 3])
    p.add_argument("--resnet_width", type=int, default=64)
    p.add_argument("--emb_vocab", type=int, default=30522)
    p.add_argument("--bucket_mb", type=float, default=64.0, help="all-reduce bucket size (data parallel)")
    p.add_argument("--overlap_adam", action="store_true", help="Adam per finished gradient range beside the backward pass (measured time-neutral; not with --clip)")


class Ctx:
    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        rehearse = bool(os.environ.get("MMVQA_REHEARSE_GLOO"))   # every rank on cuda:0 over gloo: one-GPU rehearsal only
        if rehearse:
            local = 0
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            torch.cuda.set_device(local)
            if rehearse:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        self.dev = torch.device("cuda", local if self.world > 1 else 0)
        torch.cuda.set_device(self.dev)

    def mean(self, x: float) -> float:
        if self.world == 1:
            return x
        t = torch.tensor([x], device=self.dev, dtype=torch.float64)
        dist.all_reduce(t)
        return float(t) / self.world


def build(args, ctx, n_classes=None):
    torch.manual_seed(args.seed)          # identical replicas
    model = Model(args)
    if args.backbone_weights:
        checkpoint.load_backbone(model, args.backbone_weights)
    if args.bert_weights:
        checkpoint.load_bert_embeddings(model, args.bert_weights)
    if args.state_dict:
        sd = checkpoint.read_state_dict(args.state_dict)
        own = model.state_dict()
        model.load_state_dict({k: v for k, v in sd.items() if k in own and own[k].shape == v.shape}, strict=False)
    if getattr(args, "use_pretrained", False):          # vqamed2019/train.py:125-135
        checkpoint.load_roco_pretrained(model, args.model_dir)
    if n_classes is not None:              # vqamed2019/train.py:137,141,149
        model.classifier[2] = torch.nn.Linear(args.hidden_size, n_classes)
    if getattr(args, "resume_training", False):         # vqamed2019/train.py:139-144
        checkpoint.load_model(model, args.resume_dir)
    model.to(ctx.dev)
    model.set_seed(args.seed + ctx.rank)
    cs = sync_replicas(model)              # rank 0's replica everywhere, checksum compared across ranks
    opt = FusedAdam(model, lr=args.lr)
    sched = lr_scheduler.ReduceLROnPlateau(_SchedShim(opt), patience=args.patience, factor=args.factor)
    red = GradReducer(model.flat_grads, bucket_mb=getattr(args, "bucket_mb", 64.0))
    if ctx.world > 1:
        model.set_grad_ready_hook(red.start, with_event=True)
        if ctx.rank == 0:
            print(f"data parallel: {comm_info(red)} replica checksum {cs}", flush=True)
    # opt-in: Adam per finished (all-reduced) gradient range, beside the rest of the backward pass -- not with gradient
    # clipping, which needs the norm of the WHOLE gradient before the first update (vqamed2019/utils.py:663-664)
    if getattr(args, "overlap_adam", False) and not getattr(args, "clip", False):
        opt.overlap_backward(red, grad_scale=1.0 / ctx.world)
    return model, opt, sched, red


class _SchedShim(torch.optim.Optimizer):
    """lets torch's ReduceLROnPlateau drive FusedAdam.param_groups[0]['lr']"""

    def __init__(self, fused):
        self.fused = fused
        self.param_groups = fused.param_groups
        self.defaults = {}
        self.state = {}


def save_recorder(args, epoch, model, opt, sched, mode, best=None):
    """the 5-epoch "recorder" dict of roco_train.py:164-171 / roco_supcon_train.py:177-184 (keys epoch, optimizer,
    scheduler, scaler, model), plus two keys the reference's dict lacks: which loop wrote it and the best-so-far
    trackers -- without them a resumed run re-saves its "best" checkpoint in the first epoch whatever the loss, and a
    recorder of another loop in the same save_dir is loaded blindly."""
    os.makedirs(args.save_dir, exist_ok=True)
    torch.save({"epoch": epoch, "optimizer": opt.state_dict(), "scheduler": sched.state_dict(), "scaler": {},
                "model": model.state_dict(), "mode": mode, "best": dict(best or {})},
               os.path.join(args.save_dir, "recorder_2.pt"))


def save_model(args, model, suffix=""):
    """torch.save(model.state_dict(), save_dir/task/run_name[+suffix].pt) -- roco_train.py:194-197,
    roco_supcon_train.py:199-202, vqamed2019/train.py:265-283"""
    d = os.path.join(args.save_dir, args.task)
    os.makedirs(d, exist_ok=True)
    torch.save(model.state_dict(), os.path.join(d, args.run_name + suffix + ".pt"))


def maybe_resume(args, model, opt, sched, mode):
    """-> (first epoch to run, best-so-far trackers of the interrupted run)"""
    path = os.path.join(args.save_dir, "recorder_2.pt")
    if not (args.resume and os.path.exists(path)):
        return 0, {}
    rec = torch.load(path, map_location="cpu", weights_only=False)
    if rec.get("mode", mode) != mode:
        raise RuntimeError(f"{path} was written by the '{rec['mode']}' loop, this is '{mode}': refusing to resume from it")
    model.load_state_dict(rec["model"])
    opt.load_state_dict(rec["optimizer"])
    sched.load_state_dict(rec["scheduler"])
    return rec["epoch"] + 1, dict(rec.get("best") or {})


# ----------------------------------------------------------------------------------------- one training step each
def mlm_step(model, opt, red, world, batch):
    """pretrain/roco_utils.py:214-247,257-265: zero_grad -> forward -> log_softmax + NLLLoss -> backward ->
    (gradient all-reduce) -> Adam.  Returns (loss, pred[B,T], stats = {loss, #target>0, #correct})."""
    img, ids, seg, mask, tgt = batch
    opt.zero_grad()
    loss, pred, stats = mlm_loss(model(img, ids, seg, mask), tgt)
    loss.backward()
    red.allreduce()
    opt.step(grad_scale=1.0 / world, zero_grad=True)
    return loss, pred, stats


def process_tensors(img, caption_token, aug_tokens, segment_ids, attention_mask, target, aug_targets):
    """models/SupConLoss/supcon_utils.py:253-256: the two views concatenated along the batch; segment ids and
    attention mask of view 1 are used for BOTH views"""
    cat = lambda a, b: torch.cat([a, b], dim=0)   # noqa: E731
    return (cat(img[0], img[1]), cat(caption_token, aug_tokens), cat(segment_ids, segment_ids),
            cat(attention_mask, attention_mask), cat(target, aug_targets))


def supcon_step(model, opt, red, world, batch):
    """models/SupConLoss/supcon_utils.py:270-294: MLM loss over both views + SupCon(split_feat(feat)) (called without
    a mask => SimCLR, :287); under DDP the views of all ranks are gathered first.  Returns (loss, pred, stats)."""
    img, ids, seg, mask, tgt = batch
    opt.zero_grad()
    logits, feat = model(img, ids, seg, mask)
    loss_mlm, pred, stats = mlm_loss(logits, tgt)
    bsz = img.shape[0] // 2                        # supcon_utils.py:284 (2 = n_views)
    feat = global_supcon_views(feat, bsz)          # = split_feat(feat, bsz) on one rank; global negatives under DDP
    loss = loss_mlm + supcon_loss(feat)            # 2N*world rows: the tiled kernel has no size cap
    loss.backward()
    red.allreduce()
    opt.step(grad_scale=1.0 / world, zero_grad=True)
    return loss, pred, stats


def vqa_step(model, opt, red, world, batch, crit, clip=False):
    """vqamed2019/utils.py:633-673: logits, _, _ = model(...); loss = criterion(logits, target); backward;
    optional clip_grad_norm_(1.0) (:663-664); Adam; pred = softmax(1).argmax(1)"""
    img, ids, seg, mask, tgt = batch
    opt.zero_grad()
    logits, _, _ = model(img, ids, seg, mask)       # utils.py:646
    loss = crit(logits, tgt)
    loss.backward()
    red.allreduce()
    scale = 1.0 / world
    if clip:                                        # global 2-norm over the (averaged) flat gradient buffer
        gn = float(model.flat_grads.norm()) * scale
        scale *= min(1.0, 1.0 / (gn + 1e-6))
    opt.step(grad_scale=scale, zero_grad=True)
    return loss, logits.detach().softmax(1).argmax(1)


# ----------------------------------------------------------------------------------------- MLM
def run_mlm(args):
    ctx = Ctx(args)
    args.dataset, args.task = "roco", "MLM"
    model, opt, sched, red = build(args, ctx)
    T, B, V = args.max_position_embeddings, args.batch_size, args.vocab_size
    start, kept = maybe_resume(args, model, opt, sched, "mlm")
    best = kept.get("best", float("inf"))
    for epoch in range(start, args.epochs):
        model.train()
        tl, nm, nc = 0.0, 0.0, 0.0
        for i in range(args.steps_per_epoch):
            img, ids, seg, mask, tgt = synth.roco_batch(B, T, args.image_size, min(V, args.emb_vocab),
                                                        seed=args.seed + 7919 * (epoch * 100003 + i) + ctx.rank,
                                                        device=ctx.dev, mlm_prob=args.mlm_prob)
            _, _, stats = mlm_step(model, opt, red, ctx.world, (img, ids, seg, mask, tgt))
            s = stats.tolist()               # per-step host sync, as roco_utils.py:267
            tl, nm, nc = tl + s[0], nm + s[1], nc + s[2]
        vl, va = validate_mlm(args, ctx, model, epoch)
        sched.step(vl)
        if (epoch + 1) % 5 == 0 and ctx.rank == 0:
            save_recorder(args, epoch, model, opt, sched, "mlm", {"best": min(best, vl)})
        tl = ctx.mean(tl / args.steps_per_epoch)
        if ctx.rank == 0:
            print(f"Epoch {epoch + 1}/{args.epochs} Learning rate: {opt.param_groups[0]['lr']:.7f}, Train loss: {tl:.4f}, "
                  f"Train acc: {100.0 * nc / max(nm, 1):.4f} ,Val loss: {vl:.4f}, Val acc: {va:.4f}", flush=True)
            if vl < best:
                save_model(args, model)
        best = min(best, vl)
    return best


@torch.no_grad()
def validate_mlm(args, ctx, model, epoch):
    model.eval()
    vl, nm, nc = 0.0, 0.0, 0.0
    for i in range(args.val_steps):
        img, ids, seg, mask, tgt = synth.roco_batch(args.batch_size, args.max_position_embeddings, args.image_size,
                                                    min(args.vocab_size, args.emb_vocab), seed=10 ** 6 + i + ctx.rank,
                                                    device=ctx.dev, mlm_prob=args.mlm_prob)
        out = model(img, ids, seg, mask)
        logits = out[0] if isinstance(out, tuple) else out
        _, _, stats = mlm_loss(logits, tgt)
        s = stats.tolist()
        vl, nm, nc = vl + s[0], nm + s[1], nc + s[2]
    return ctx.mean(vl / args.val_steps), 100.0 * nc / max(nm, 1)


# ----------------------------------------------------------------------------------------- MLM + SupCon
def run_supcon(args):
    ctx = Ctx(args)
    args.dataset, args.task, args.supcon = "roco", "MLM", True
    model, opt, sched, red = build(args, ctx)
    T, V = args.max_position_embeddings, args.vocab_size
    n = args.batch_size // 2                      # roco_supcon_train.py:137: the loader yields bs//2 pairs
    if n < 1:
        raise ValueError("--batch_size must be >= 2 (two views per sample)")
    start, kept = maybe_resume(args, model, opt, sched, "supcon")
    best = kept.get("best", float("inf"))
    for epoch in range(start, args.epochs):
        model.train()
        tl = 0.0
        for i in range(args.steps_per_epoch):
            sd = args.seed + 7919 * (epoch * 100003 + i) + ctx.rank
            a = synth.roco_batch(n, T, args.image_size, min(V, args.emb_vocab), seed=sd, device=ctx.dev, mlm_prob=args.mlm_prob)
            b = synth.roco_batch(n, T, args.image_size, min(V, args.emb_vocab), seed=sd + 1, device=ctx.dev, mlm_prob=args.mlm_prob)
            batch = process_tensors((a[0], b[0]), a[1], b[1], a[2], a[3], a[4], b[4])
            loss, _, _ = supcon_step(model, opt, red, ctx.world, batch)
            tl += float(loss.detach())
        vl, va = validate_mlm(args, ctx, model, epoch)
        sched.step(vl)
        if (epoch + 1) % 5 == 0 and ctx.rank == 0:       # roco_supcon_train.py:177-184
            save_recorder(args, epoch, model, opt, sched, "supcon", {"best": min(best, vl)})
        if ctx.rank == 0:
            print(f"Epoch {epoch + 1}/{args.epochs} Learning rate: {opt.param_groups[0]['lr']:.7f}, "
                  f"Train loss: {tl / args.steps_per_epoch:.4f}, Val loss: {vl:.4f}, Val acc: {va:.4f}", flush=True)
            if vl < best:                                 # roco_supcon_train.py:199-202
                save_model(args, model)
        best = min(best, vl)
    return best


# ----------------------------------------------------------------------------------------- VQA-Med-2019
def run_vqa(args):
    ctx = Ctx(args)
    args.dataset, args.task = "VQA-Med", "MLM"
    C = args.num_classes
    model, opt, sched, red = build(args, ctx, n_classes=C)
    T, B = args.max_position_embeddings, args.batch_size
    crit = (lambda lg, t: asl_loss(lg, t)) if args.loss == "ASLSingleLabel" else (lambda lg, t: mlm_loss(lg, t)[0])
    # (vqamed2019/train.py itself has no recorder / --resume; kept here like the two pre-training loops)
    start, kept = maybe_resume(args, model, opt, sched, "vqa")
    best_loss, best_acc1 = kept.get("best_loss", float("inf")), kept.get("best_acc1", 0.0)
    best_acc2, counter = kept.get("best_acc2", 0.0), kept.get("counter", 0)
    for epoch in range(start, args.epochs):
        model.train()
        tl = 0.0
        for i in range(args.steps_per_epoch):
            img, ids, seg, mask, tgt = synth.vqa_batch(B, T, args.image_size, args.emb_vocab, C,
                                                       seed=args.seed + 7919 * (epoch * 100003 + i) + ctx.rank, device=ctx.dev)
            loss, _ = vqa_step(model, opt, red, ctx.world, (img, ids, seg, mask, tgt), crit, clip=args.clip)
            tl += float(loss.detach())
        model.eval()
        vl, correct, total = 0.0, 0, 0
        with torch.no_grad():
            for i in range(args.val_steps):
                img, ids, seg, mask, tgt = synth.vqa_batch(B, T, args.image_size, args.emb_vocab, C, seed=10 ** 6 + i, device=ctx.dev)
                logits, _, _ = model(img, ids, seg, mask)
                vl += float(crit(logits, tgt))
                correct += int((logits.softmax(1).argmax(1) == tgt).sum())   # utils.py:673
                total += B
        vl, acc = ctx.mean(vl / args.val_steps), 100.0 * correct / total
        sched.step(vl)
        if ctx.rank == 0:
            print(f"Epoch {epoch + 1}/{args.epochs} lr {opt.param_groups[0]['lr']:.7f} train_loss {tl / args.steps_per_epoch:.4f} "
                  f"val_loss {vl:.4f} val_total_acc {acc:.2f}", flush=True)
        if ctx.rank == 0:
            if vl < best_loss:                   # train.py:264-268 "save by val loss"
                save_model(args, model, "_loss")
            if acc > best_acc1:                  # train.py:270-276 "save by accuracy in val"
                save_model(args, model)
        best_loss = min(best_loss, vl)
        best_acc1 = max(best_acc1, acc)
        expired = False
        if best_acc1 > best_acc2:                # train.py:288-296 early stop
            counter, best_acc2 = 0, best_acc1
        else:
            counter += 1
            expired = counter > args.counter
        if (epoch + 1) % 5 == 0 and ctx.rank == 0:
            save_recorder(args, epoch, model, opt, sched, "vqa",
                          dict(best_loss=best_loss, best_acc1=best_acc1, best_acc2=best_acc2, counter=counter))
        if expired:
            if ctx.rank == 0:
                print("Counter expired, finishing.")
            break
    return best_loss


# ----------------------------------------------------------------------------------------- VQA-Med-2019 test-set run
def run_eval(args):
    """vqamed2019/eval.py:99-180: Model(args) -> classifier[2] = Linear(hidden, num_classes) -> load_state_dict(model_dir)
    -> test() over the test split (batch_size, shuffle False) -> print acc / bleu -> <model_name>_preds.csv and
    <model_name>_res.txt in save_dir.  The test split is synthetic (mmvqa_amd.synth.vqa_test_table + vqa_batch: the
    dataset and its tokenizer are not in the image); everything after the loader is the reference's sequence."""
    ctx = Ctx(args)
    args.dataset, args.task = "VQA-Med", "MLM"
    C = args.num_classes
    torch.manual_seed(args.seed)
    model = Model(args)
    model.classifier[2] = torch.nn.Linear(args.hidden_size, C)                     # eval.py:109
    if args.model_dir:
        print("Loading model at ", args.model_dir)
        model.load_state_dict(checkpoint.read_state_dict(args.model_dir))        # eval.py:112
    model.to(ctx.dev)
    crit = (lambda lg, t: asl_loss(lg, t)) if args.loss == "ASLSingleLabel" else (lambda lg, t: mlm_loss(lg, t)[0])
    cols, rows, idx2ans = synth.vqa_test_table(args.test_samples, C, seed=args.seed)
    B, T = args.batch_size, args.max_position_embeddings

    def loader():                                                                # DataLoader(testdataset, batch_size, shuffle=False)
        for lo in range(0, len(rows), B):
            n = min(B, len(rows) - lo)
            img, ids, seg, mask, _ = synth.vqa_batch(n, T, args.image_size, args.emb_vocab, C, seed=args.seed + lo, device=ctx.dev)
            tgt = torch.tensor([r[2] for r in rows[lo:lo + n]], dtype=torch.long, device=ctx.dev)
            yield img, ids, seg, mask, tgt

    cats = [r[3] for r in rows]
    test_loss, predictions, acc, bleu = evaluate.test(loader(), model, crit, cats, idx2ans, category=args.category)
    model_name = (args.model_dir or args.run_name).split("/")[-1]               # eval.py:68
    if ctx.rank == 0:
        paths = evaluate.write_test_files(rows, cols, predictions, idx2ans, args.save_dir, model_name)   # eval.py:171-178
        print("test_loss", float(test_loss))
        print("acc", acc)
        print("bleu", bleu)
        print("wrote", *paths)
    return test_loss, acc, bleu


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    mode = argv.pop(0) if argv and argv[0] in ("mlm", "supcon", "vqa", "eval") else "mlm"
    p = argparse.ArgumentParser(description=f"mmvqa_amd training ({mode})")
    common_args(p)
    if mode in ("mlm", "supcon"):
        p.add_argument("--mlm_prob", type=float, default=0.15)
        p.add_argument("--lr", type=float, default=2e-5)
        p.add_argument("--max_position_embeddings", type=int, default=75)
        if mode == "supcon":
            p.add_argument("--con_task", type=str, default="supcon", choices=["simclr", "supcon"])
            p.add_argument("--similarity", type=str, default="sentence_transformers")
    else:
        p.add_argument("--lr", type=float, default=1e-4)
        p.add_argument("--max_position_embeddings", type=int, default=28)
        p.add_argument("--loss", type=str, default="CrossEntropyLoss", choices=["CrossEntropyLoss", "ASLSingleLabel"])
        p.add_argument("--num_classes", type=int, default=1552)
        p.add_argument("--counter", type=int, default=20)
        p.add_argument("--clip", action="store_true", default=False, help="clip_grad_norm_(1.0), utils.py:663-664")
        p.add_argument("--use_pretrained", action="store_true", default=False)      # vqamed2019/train.py:69-72
        p.add_argument("--model_dir", type=str, default=None, help="ROCO-pretrained Model state_dict")
        p.add_argument("--resume_training", action="store_true", default=False)
        p.add_argument("--resume_dir", type=str, default=None, help="fine-tuned Model state_dict to continue from")
        p.add_argument("--category", type=str, default=None, help="eval: one question category only (eval.py:29)")
        p.add_argument("--test_samples", type=int, default=64, help="eval: size of the synthetic test split")
    args = p.parse_args(argv)
    out = {"mlm": run_mlm, "supcon": run_supcon, "vqa": run_vqa, "eval": run_eval}[mode](args)
    if dist.is_initialized():
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
