"""GPU smoke tests of the training loops (callers of the hot path): a few synthetic steps of each of
the three loops with a reduced backbone.  The loss must DECREASE by a margin on a fixed seed (the initial loss is
ln(classes)), the reference's checkpoint files must appear, and --resume must carry the optimizer state AND keep the
learning-rate scheduler wired to the fused Adam."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

MINI = ["--resnet_layers", "1", "1", "1", "1", "--resnet_width", "8", "--hidden_size", "96", "--n_layers", "2",
        "--vocab_size", "64", "--emb_vocab", "64", "--image_size", "32", "--steps_per_epoch", "6", "--val_steps", "2",
        "--epochs", "5", "--max_position_embeddings", "16", "--hidden_dropout_prob", "0.1"]


def test_mlm_loop(tmp_path):
    from mmvqa_amd import train
    best = train.main(["mlm", "--lr", "1e-3", "--batch_size", "4", "--save_dir", str(tmp_path)] + MINI)
    assert best == best and best < math.log(64) - 0.25          # ln(64) = 4.16 at init: it must learn
    assert (tmp_path / "MLM" / "run.pt").exists()               # roco_train.py:194-197
    assert (tmp_path / "recorder_2.pt").exists()                # roco_train.py:164-171 (every 5 epochs)
    rec = torch.load(tmp_path / "recorder_2.pt", weights_only=False)
    # the reference's five keys + which loop wrote it and its best-so-far trackers (needed for a correct --resume)
    assert set(rec) == {"epoch", "optimizer", "scheduler", "scaler", "model", "mode", "best"} and rec["epoch"] == 4
    assert rec["mode"] == "mlm" and rec["best"]["best"] == pytest.approx(best)


def test_supcon_loop(tmp_path):
    from mmvqa_amd import train
    best = train.main(["supcon", "--lr", "1e-3", "--batch_size", "8", "--transformer_model", "realformer",
                       "--save_dir", str(tmp_path)] + MINI)
    assert best == best and best < math.log(64) - 0.25
    assert (tmp_path / "MLM" / "run.pt").exists()               # roco_supcon_train.py:199-202
    assert (tmp_path / "recorder_2.pt").exists()                # roco_supcon_train.py:177-184


def test_vqa_loop_asl_and_ce(tmp_path):
    from mmvqa_amd import train
    for loss in ("ASLSingleLabel", "CrossEntropyLoss"):
        d = tmp_path / loss
        best = train.main(["vqa", "--lr", "1e-3", "--batch_size", "8", "--loss", loss, "--num_classes", "11",
                           "--save_dir", str(d)] + MINI)
        assert best == best and best < 10.0
        assert (d / "MLM" / "run_loss.pt").exists() and (d / "MLM" / "run.pt").exists()   # train.py:264-276


def test_resume_keeps_scheduler_wired(tmp_path):
    """--resume: the recorder's optimizer state is restored IN PLACE, so a later ReduceLROnPlateau reduction still
    reaches mmvqa_adam (the lr the kernel gets is FusedAdam.param_groups[0]['lr'])"""
    import argparse
    import mmvqa_amd
    from mmvqa_amd import train
    from oracle import mmbert_oracle as O
    train.main(["mlm", "--lr", "1e-3", "--batch_size", "4", "--save_dir", str(tmp_path)] + MINI)
    args = O.make_args(resnet_layers=(1, 1, 1, 1), resnet_width=8, hidden_size=96, n_layers=2, vocab_size=64, emb_vocab=64)
    model = mmvqa_amd.Model(args).to("cuda")
    opt = mmvqa_amd.FusedAdam(model, lr=5e-2)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(train._SchedShim(opt), patience=0, factor=0.1)
    ns = argparse.Namespace(save_dir=str(tmp_path), resume=True)
    start, kept = train.maybe_resume(ns, model, opt, sched, "mlm")
    assert start == 5 and opt.step_count == 30
    assert kept["best"] < 10.0                                  # the best validation loss travels with the recorder
    with pytest.raises(RuntimeError, match="written by the 'mlm' loop"):
        train.maybe_resume(ns, model, opt, sched, "vqa")        # another loop's recorder in the same save_dir is refused
    assert opt.param_groups[0]["lr"] == pytest.approx(1e-3)     # the checkpoint's lr, not the constructor's
    assert float(opt.m.abs().sum()) > 0
    for _ in range(sched.patience + 1):                         # (patience/best come from the checkpoint)
        sched.step(1e9)                                         # no improvement for patience+1 epochs -> reduce
    assert opt.param_groups[0]["lr"] == pytest.approx(1e-4)     # ... and the fused optimizer sees it
    p0 = model.flat_params.clone()
    model.flat_grads.fill_(1.0)
    opt.step()
    step = float((p0 - model.flat_params).abs().max())
    assert 0 < step <= 1.2e-4                                   # |dp| <= lr for Adam


def test_eval_subcommand_writes_the_two_files(tmp_path):
    """vqamed2019/eval.py:99-180 counterpart: checkpoint of a fine-tuning run -> test-set metrics + the two files"""
    import csv
    from mmvqa_amd import train
    d = tmp_path / "ft"
    train.main(["vqa", "--lr", "1e-3", "--batch_size", "8", "--loss", "ASLSingleLabel", "--num_classes", "11",
                "--save_dir", str(d)] + MINI)
    out = tmp_path / "out"
    loss, acc, bleu = train.main(["eval", "--model_dir", str(d / "MLM" / "run.pt"), "--num_classes", "11", "--batch_size", "8",
                                  "--test_samples", "20", "--save_dir", str(out)] + MINI)
    assert loss == loss
    assert set(acc) == {"total_acc", "binary_acc", "plane_acc", "organ_acc", "modality_acc", "abnorm_acc"}
    assert set(bleu) == {k.replace("_acc", "_bleu") for k in acc}
    rows = list(csv.reader(open(out / "run.pt_preds.csv")))
    assert rows[0] == ["img_id", "question", "answer", "category", "mode", "preds", "decode_preds", "decode_ans"]
    assert len(rows) == 21
    res = open(out / "run.pt_res.txt").read().splitlines()
    assert len(res) == 20 and res[0].startswith("synpic10000|")
