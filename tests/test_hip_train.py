"""GPU smoke tests of the training loops (callers of the hot path): a few synthetic steps of each of
the three loops with a reduced backbone; the loss must be finite and decrease on a fixed seed."""
import pytest
import torch

pytestmark = pytest.mark.gpu

MINI = ["--resnet_layers", "1", "1", "1", "1", "--resnet_width", "8", "--hidden_size", "96", "--n_layers", "2",
        "--vocab_size", "64", "--emb_vocab", "64", "--image_size", "32", "--steps_per_epoch", "6", "--val_steps", "2",
        "--epochs", "3", "--max_position_embeddings", "16", "--hidden_dropout_prob", "0.1"]


def test_mlm_loop(tmp_path):
    from mmvqa_amd import train
    best = train.main(["mlm", "--lr", "1e-3", "--batch_size", "4", "--save_dir", str(tmp_path)] + MINI)
    assert best == best and best < 4.2          # ln(64) = 4.16 at init
    assert (tmp_path / "MLM" / "run.pt").exists()


def test_supcon_loop(tmp_path):
    from mmvqa_amd import train
    best = train.main(["supcon", "--lr", "1e-3", "--batch_size", "8", "--transformer_model", "realformer",
                       "--save_dir", str(tmp_path)] + MINI)
    assert best == best and best < 4.2


def test_vqa_loop_asl_and_ce(tmp_path):
    from mmvqa_amd import train
    for loss in ("ASLSingleLabel", "CrossEntropyLoss"):
        best = train.main(["vqa", "--lr", "1e-3", "--batch_size", "8", "--loss", loss, "--num_classes", "11",
                           "--save_dir", str(tmp_path)] + MINI)
        assert best == best and best < 10.0
