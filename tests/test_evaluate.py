"""Evaluation metrics of the VQA-Med loop (vqamed2019/utils.py:328-330, 740-765, 813-841).
CPU part: BLEU-1 known answers (nltk is absent, so the restatement is anchored by hand-computed values) and the
per-category bookkeeping against the oracle restatement; GPU part: validate()/test() on the HIP model vs the oracle."""
import math

import numpy as np
import pytest
import torch

from mmvqa_amd import evaluate as EV
from oracle import loops_oracle as LO


def test_bleu1_known_answers():
    b = EV.sentence_bleu1
    assert b("the cat sat".split(), "the cat sat".split()) == pytest.approx(1.0)
    assert b("yes".split(), "no".split()) == 0.0
    assert b("axial".split(), "axial".split()) == pytest.approx(1.0)
    # hypothesis shorter than the reference: precision 1, brevity penalty exp(1 - 3/1)
    assert b("ct with contrast".split(), "ct".split()) == pytest.approx(math.exp(-2.0))
    # hypothesis longer: 2 of 4 unigrams match, no penalty
    assert b("mr flair".split(), "mr t2 weighted flair".split()) == pytest.approx(0.5)
    # clipping: 'the' counted at most as often as in the reference
    assert b("the cat".split(), "the the the".split()) == pytest.approx(1.0 / 3.0)
    # same length, one of two words right
    assert b("left lung".split(), "right lung".split()) == pytest.approx(0.5)
    assert b("a b".split(), []) == 0.0
    for ref, hyp in (("a b c d", "a c"), ("x", "x y z"), ("p q", "q p"), ("m n o", "m")):
        assert b(ref.split(), hyp.split()) == pytest.approx(LO.sentence_bleu_unigram([ref.split()], hyp.split()))


def test_category_metrics_match_oracle_bookkeeping():
    rng = np.random.default_rng(0)
    idx2ans = {0: "yes", 1: "no", 2: "axial", 3: "coronal", 4: "lung", 5: "ct with contrast", 6: "ct", 7: "mr flair",
               8: "pulmonary embolism", 9: "embolism"}
    n = 200
    targets = rng.integers(0, 10, n)
    preds = np.where(rng.random(n) < 0.6, targets, rng.integers(0, 10, n))
    cats = rng.choice(["binary", "plane", "organ", "modality", "abnormality"], n)

    class FakeModel(torch.nn.Module):          # emits logits whose argmax is the wanted prediction
        def __init__(self):
            super().__init__()
            self.i = 0

        def forward(self, img, q, s, m):
            B = img.shape[0]
            lg = torch.zeros(B, 10)
            lg[torch.arange(B), torch.from_numpy(preds[self.i:self.i + B])] = 5.0
            self.i += B
            return lg, 0, 0

    loader = [(torch.zeros(20, 1), None, None, None, torch.from_numpy(targets[i:i + 20])) for i in range(0, n, 20)]
    crit = torch.nn.CrossEntropyLoss()
    for fn, prefix in ((EV.validate, "val_"), (EV.test, "")):
        loss, P, acc, bleu = fn(loader, FakeModel(), crit, cats, idx2ans)
        loss_o, P_o, acc_o, bleu_o = LO.vqa_validate(loader, FakeModel(), crit, cats, idx2ans, prefix)
        assert np.array_equal(P, P_o) and np.array_equal(P, preds)
        assert float(loss) == pytest.approx(float(loss_o))
        assert acc.keys() == acc_o.keys() and bleu.keys() == bleu_o.keys()
        assert set(acc) == {prefix + k for k in ("total_acc", "binary_acc", "plane_acc", "organ_acc", "modality_acc", "abnorm_acc")}
        for k in acc:
            assert acc[k] == acc_o[k], k
        for k in bleu:
            assert bleu[k] == bleu_o[k], k
    # --category form (utils.py:741-743): plain numbers
    loss, P, acc, bleu = EV.validate(loader, FakeModel(), crit, cats, idx2ans, category="organ")
    assert acc == pytest.approx((preds == targets).mean() * 100.) and 0.0 <= bleu <= 1.0


def load_eval_fixture(golden_dir):
    import os
    g = dict(np.load(os.path.join(golden_dir, "loop_vqa_eval.npz"), allow_pickle=False))
    B, T, hw, C, nb = [int(v) for v in g["dims"]]
    loader = [tuple(torch.from_numpy(g[f"{n}{i}"]) for n in ("img", "ids", "seg", "mask", "tgt")) for i in range(nb)]
    idx2ans = {i: str(a) for i, a in enumerate(g["answers"])}
    return g, loader, idx2ans, C


def eval_fixture_oracle_model(g, C):
    from oracle import mmbert_oracle as O
    args = O.make_args(transformer_model="realformer", dataset="VQA-Med", hidden_size=768, n_layers=2, heads=12,
                       hidden_dropout_prob=0.0, vocab_size=C, emb_vocab=C, resnet_layers=(1, 1, 1, 1), resnet_width=64,
                       bert_max_pos=32, use_relu=False, cnn_encoder="resnet152")
    torch.manual_seed(int(g["seed"]))
    orc = O.OracleModel(args)
    LO.perturb_bn_buffers(orc, seed=int(g["bn_seed"]))
    return args, orc


def same_metrics(got, keys, vals):
    assert list(got.keys()) == [str(k) for k in keys]
    for k, v in zip(keys, vals):
        a = float(got[str(k)])
        assert (math.isnan(a) and math.isnan(v)) or a == v, (k, a, v)


def test_oracle_validate_and_test_match_the_reference_functions(golden_dir):
    """oracle/loops_oracle.vqa_validate against tests/golden/loop_vqa_eval.npz = the reference's OWN validate / test
    (vqamed2019/utils.py:690-843) on a loader with mixed categories, one of them empty"""
    from oracle import mmbert_oracle as O
    g, loader, idx2ans, C = load_eval_fixture(golden_dir)
    _, orc = eval_fixture_oracle_model(g, C)
    cats = g["categories"]
    assert "organ" not in set(cats)
    for name, prefix, crit in (("val", "val_", O.asl_single_label), ("test", "", torch.nn.CrossEntropyLoss())):
        loss, P, acc, bleu = LO.vqa_validate(loader, orc, crit, cats, idx2ans, prefix)
        assert np.array_equal(P, g[f"{name}_preds"])
        assert abs(float(loss) - float(g[f"{name}_loss"])) <= 2e-5 * abs(float(g[f"{name}_loss"]))
        same_metrics(acc, g[f"{name}_acc_keys"], g[f"{name}_acc_vals"])
        same_metrics(bleu, g[f"{name}_bleu_keys"], g[f"{name}_bleu_vals"])
        assert math.isnan(float(acc[prefix + "organ_acc"])) and math.isnan(float(bleu[prefix + "organ_bleu"]))
    loss, P, acc, bleu = LO.vqa_validate(loader, orc, O.asl_single_label, cats, idx2ans, "val_", category="plane")
    assert acc == float(g["cat_acc"]) and bleu == float(g["cat_bleu"])
    # the product's host bookkeeping on the reference's predictions
    t = np.concatenate([b[4].numpy() for b in loader])
    acc, bleu = EV.category_metrics(g["val_preds"], t, cats, idx2ans, "val_")
    same_metrics(acc, g["val_acc_keys"], g["val_acc_vals"])
    same_metrics(bleu, g["val_bleu_keys"], g["val_bleu_vals"])


def test_test_files_are_byte_equal_to_the_pandas_ones(tmp_path):
    """eval.py:171-178 writes the two files with DataFrame.to_csv; the build writes them with the csv module: same bytes,
    including fields that need quoting (commas, quotes, the '|' separator of the result file)"""
    import pandas as pd
    idx2ans = {0: "yes", 1: "no", 2: "ct, with contrast", 3: 'the "left" lung', 4: "mr | flair", 5: "axial"}
    cols = ["img_id", "question", "answer", "category", "mode"]
    rows = [("../ImageClef-2019-VQA-Med/Test/images/synpic%d.jpg" % (100 + i), q, a, c, "test")
            for i, (q, a, c) in enumerate([("is this a ct?", 0, "binary"), ("what plane, exactly?", 5, "plane"),
                                           ('what is "abnormal" here?', 3, "abnormality"), ("modality?", 2, "modality"),
                                           ("which | organ", 4, "organ"), ("plain", 1, "binary")])]
    preds = np.array([1, 5, 2, 2, 4, 3])
    a, b = tmp_path / "build", tmp_path / "pandas"
    b.mkdir()
    EV.write_test_files(rows, cols, preds, idx2ans, str(a), "m.pt")
    LO.eval_write_csvs(pd.DataFrame(rows, columns=cols), preds, idx2ans, str(b), "m.pt")
    for f in ("m.pt_preds.csv", "m.pt_res.txt"):
        assert (a / f).read_bytes() == (b / f).read_bytes(), f
    assert (a / "m.pt_res.txt").read_text().splitlines()[0] == "synpic100|no"


@pytest.mark.gpu
def test_reference_validate_and_test_fixture_replays_on_the_hip_model(golden_dir):
    """the same fixture through mmvqa_amd.evaluate.validate / test with the HIP engine doing the eval-mode forward:
    predictions bit-exact, loss within 1e-3, every accuracy / BLEU value (and the nan of the empty category) equal"""
    import mmvqa_amd
    from hip_helpers import dev
    g, loader, idx2ans, C = load_eval_fixture(golden_dir)
    args, orc = eval_fixture_oracle_model(g, C)
    hip = mmvqa_amd.Model(args)
    hip.load_state_dict(orc.state_dict())
    hip.to(dev())
    gl = [tuple(t.to(dev()) for t in b) for b in loader]
    cats = g["categories"]
    for name, fn, crit in (("val", EV.validate, mmvqa_amd.asl_loss), ("test", EV.test, lambda lg, t: mmvqa_amd.mlm_loss(lg, t)[0])):
        loss, P, acc, bleu = fn(gl, hip, crit, cats, idx2ans)
        assert np.array_equal(P, g[f"{name}_preds"])
        assert abs(float(loss) - float(g[f"{name}_loss"])) <= 1e-3 * abs(float(g[f"{name}_loss"]))
        same_metrics(acc, g[f"{name}_acc_keys"], g[f"{name}_acc_vals"])
        same_metrics(bleu, g[f"{name}_bleu_keys"], g[f"{name}_bleu_vals"])
    loss, P, acc, bleu = EV.validate(gl, hip, mmvqa_amd.asl_loss, cats, idx2ans, category="plane")
    assert acc == float(g["cat_acc"]) and bleu == float(g["cat_bleu"])
    assert not hip.training


@pytest.mark.gpu
def test_validate_on_the_hip_model_matches_oracle():
    import mmvqa_amd
    from mmvqa_amd import synth
    from oracle import mmbert_oracle as O
    from hip_helpers import dev
    from test_hip_model import build_pair, mini_args
    C = 23
    args = mini_args(transformer_model="realformer", dataset="VQA-Med", vocab_size=C)
    orc, hip = build_pair(args, seed=31)
    idx2ans = {i: " ".join(["w%d" % (i % 5), "v%d" % (i % 3)][: 1 + i % 2]) for i in range(C)}
    cpu_loader = [synth.vqa_batch(4, 10, 32, vocab=50, n_classes=C, seed=60 + i)[:5] for i in range(3)]
    cats = np.array(["binary", "plane", "organ", "modality", "abnormality", "plane"] * 2)
    gpu_loader = [tuple(t.to(dev()) for t in b) for b in cpu_loader]
    for crit_o, crit_h in ((O.asl_single_label, mmvqa_amd.asl_loss),
                           (torch.nn.CrossEntropyLoss(), lambda lg, t: mmvqa_amd.mlm_loss(lg, t)[0])):
        loss_o, P_o, acc_o, bleu_o = LO.vqa_validate(cpu_loader, orc, crit_o, cats, idx2ans)
        loss, P, acc, bleu = EV.validate(gpu_loader, hip, crit_h, cats, idx2ans)
        assert np.array_equal(P, P_o)
        assert abs(float(loss) - float(loss_o)) <= 1e-3 * abs(float(loss_o))
        assert acc == acc_o and bleu == bleu_o
    assert not hip.training          # validate() leaves the model in eval mode, as the reference does
