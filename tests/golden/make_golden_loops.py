#!/usr/bin/env python3
"""Golden vectors for the step loops, produced by running the REFERENCE's own
`train_one_epoch` functions on the reference's own Model:

    pretrain/roco_utils.py:207-290            (MLM)
    models/SupConLoss/supcon_utils.py:263-323 (MLM + SupCon, con_task 'simclr' -> buildMask returns None, :195-199)
    vqamed2019/utils.py:625-688               (VQA-Med, ASLSingleLabel)

Run ONCE in the build container:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_loops.py

Packages the image lacks are NAME-ONLY stubs (wandb, nltk, pytorch_lightning, sentence_transformers, googletrans,
bert_score, torchvision, timm); `sentence_bleu` is a stub returning 0.0 in the training loops (the VQA loop computes a BLEU
figure at the end of the epoch which those fixtures do not record) and the build's own unigram BLEU
(oracle.loops_oracle.sentence_bleu_unigram) in the evaluation fixture `loop_vqa_eval`, which runs the reference's own
`validate` and `test` (vqamed2019/utils.py:690-843) on a loader with mixed categories -- one category empty -- so that the
per-category bookkeeping, the NaN of an empty category, the rounding and the key names are pinned.  Model weights are
seeded oracle weights copied into the reference Model (state_dict names are identical), optimizer = torch.optim.Adam as the scripts build it.  Stored: the batches,
the per-step losses (recorded by wrapping the criterion objects), the returned (mean loss, accuracy) and samples of
the parameters AFTER the two optimizer steps.  Data only.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from transformers import BertTokenizer, BertModel, AutoTokenizer, AutoModel  # noqa: E402,F401
import make_golden as MG  # noqa: E402  (stubs torchvision/timm, imports the reference model modules)
from make_golden_text import stub, register_stubs  # noqa: E402
from oracle import mmbert_oracle as O  # noqa: E402
from oracle import loops_oracle as LO  # noqa: E402
from mmvqa_amd import synth  # noqa: E402

register_stubs()
sys.modules["nltk.translate.bleu_score"].sentence_bleu = lambda *a, **k: 0.0
sys.modules["torchvision"].models = sys.modules["torchvision.models"]
stub("sentence_transformers", SentenceTransformer=object, util=object)
stub("googletrans", Translator=object)
stub("bert_score", BERTScorer=object)
# make_golden_text.register_stubs() replaced torchvision.models: put the callables the reference dereferences back
sys.modules["torchvision.models"].resnet152 = lambda **kw: None
sys.path.insert(0, "/root/reference/pretrain")
sys.path.insert(0, "/root/reference/vqamed2019")
sys.path.insert(0, "/root/reference/models/SupConLoss")
import importlib  # noqa: E402

RU = importlib.import_module("roco_utils")          # (supcon_utils does `from roco_utils import encode_text`)
SU = importlib.import_module("supcon_utils")
VU = importlib.import_module("vqamed2019.utils")
from models.SupConLoss.loss import SupConLoss  # noqa: E402
from models.asl_singlelabel import ASLSingleLabel  # noqa: E402

LR = 1e-3
PICK_COMMON = ["fc1.weight", "classifier.0.bias", "classifier.1.weight", "classifier.2.weight",
               "transformer.bert_embedding.word_embeddings.weight", "transformer.bert_embedding.LayerNorm.bias",
               "transformer.trans.conv2.weight", "transformer.trans.conv7.weight",
               "transformer.trans.model.conv1.weight", "transformer.trans.model.bn1.bias",
               "transformer.trans.model.layer2.0.conv2.weight", "transformer.trans.model.layer4.0.bn3.weight"]


class Recorder(torch.nn.Module):
    """wraps a criterion and keeps every value it returned"""

    def __init__(self, fn):
        super().__init__()
        self.fn, self.values = fn, []

    def forward(self, *a, **k):
        v = self.fn(*a, **k)
        self.values.append(float(v.detach()))
        return v


def sample(t):
    t = t.detach().flatten()
    return t if t.numel() <= 4096 else t[:: max(1, t.numel() // 4096)][:4096]


def build(tm, dataset, supcon, V):
    kw = dict(transformer_model=tm, dataset=dataset, hidden_size=768, n_layers=2, heads=12, hidden_dropout_prob=0.0,
              vocab_size=V, resnet_layers=(1, 1, 1, 1), resnet_width=64, bert_max_pos=32, use_relu=False,
              cnn_encoder="resnet152")
    if supcon:
        kw["supcon"] = True
    args = O.make_args(**kw)
    torch.manual_seed(61)
    orc = O.OracleModel(args)
    ref = MG.build_ref_model(args, orc)
    MG.zero_dropout(ref)
    return ref, MG.wsum(orc.state_dict())


def finish(tag, ref, extra, pick_extra):
    sd = dict(ref.named_parameters())
    arrs = dict(extra)
    for k in PICK_COMMON + pick_extra:
        arrs["p_" + k.replace(".", "__")] = sample(sd[k])
    bsd = ref.state_dict()
    for k in ("transformer.trans.model.bn1.running_mean", "transformer.trans.model.bn1.num_batches_tracked",
              "transformer.trans.model.layer3.0.bn2.running_var"):
        arrs["b_" + k.replace(".", "__")] = bsd[k]
    MG.save(tag, **arrs)


def loop_mlm():
    V, B, T, hw = 64, 3, 12, 64
    ref, ws = build("transformer", "roco", False, V)
    batches = [synth.roco_batch(B, T, hw, vocab=V, seed=70 + i, mlm_prob=0.4) for i in range(2)]
    loader = [(img, ids.unsqueeze(1), seg, mask.unsqueeze(1), tgt) for img, ids, seg, mask, tgt in batches]   # :218-219 squeeze(1)
    crit = Recorder(torch.nn.NLLLoss())                                      # roco_train.py:89
    opt = torch.optim.Adam(ref.parameters(), lr=LR)                          # roco_train.py:90
    a = types.SimpleNamespace(mixed_precision=False, task="MLM")
    mean_loss, total_acc = RU.train_one_epoch(loader, ref, crit, opt, None, "cpu", a, 0)
    extra = dict(seed=61, wsum=ws, dims=[B, T, hw, V], lr=LR, losses=np.array(crit.values), mean_loss=mean_loss,
                 total_acc=total_acc)
    for i, b in enumerate(batches):
        for n, t in zip(("img", "ids", "seg", "mask", "tgt"), b):
            extra[f"{n}{i}"] = t
    finish("loop_mlm", ref, extra, ["transformer.blocks.norm1.weight", "transformer.blocks.attention.1.proj_v.weight"])


def loop_supcon():
    V, n, T, hw = 64, 2, 11, 64
    ref, ws = build("realformer", "roco", True, V)
    items, extra = [], {}
    for i in range(2):
        a = synth.roco_batch(n, T, hw, vocab=V, seed=80 + 2 * i, mlm_prob=0.4)
        b = synth.roco_batch(n, T, hw, vocab=V, seed=81 + 2 * i, mlm_prob=0.4)
        # ROCO_SupCon item (supcon_utils.py:270): (img[2], caption_token, aug_tokens, segment_ids, attention_mask, target, aug_targets, caption_text, aug_text)
        items.append(([a[0], b[0]], a[1].unsqueeze(1), b[1].unsqueeze(1), a[2], a[3].unsqueeze(1), a[4], b[4], None, None))
        for nm, t in (("img_a", a[0]), ("img_b", b[0]), ("ids_a", a[1]), ("ids_b", b[1]), ("seg", a[2]), ("mask", a[3]),
                      ("tgt_a", a[4]), ("tgt_b", b[4])):
            extra[f"{nm}{i}"] = t
    crit, sc = Recorder(torch.nn.NLLLoss()), Recorder(SupConLoss(temperature=0.07))   # roco_supcon_train.py:103-104
    opt = torch.optim.Adam(ref.parameters(), lr=LR)
    a = types.SimpleNamespace(con_task="simclr")
    mean_loss, total_acc = SU.train_one_epoch(items, ref, crit, sc, opt, "cpu", a, 0, None)
    extra.update(seed=61, wsum=ws, dims=[n, T, hw, V], lr=LR, losses_mlm=np.array(crit.values),
                 losses_supcon=np.array(sc.values), mean_loss=mean_loss, total_acc=total_acc)
    finish("loop_supcon", ref, extra, ["head.0.weight", "head.2.bias", "transformer.mains.0.kqv.weight",
                                       "transformer.mains.1.ln2.weight"])


def loop_vqa():
    C, B, T, hw = 23, 4, 10, 64
    ref, ws = build("realformer", "VQA-Med", False, C)
    batches = [synth.vqa_batch(B, T, hw, vocab=C, n_classes=C, seed=90 + i) for i in range(2)]
    # VQAMed train item (utils.py:249-253): (img, tokens, segment_ids, input_mask, answer, path, category)
    loader = [(img, ids.unsqueeze(1), seg, mask.unsqueeze(1), tgt, ["p"] * B, torch.zeros(B, dtype=torch.long))
              for img, ids, seg, mask, tgt in batches]
    crit = Recorder(ASLSingleLabel())                                         # vqamed2019/train.py:172-174
    opt = torch.optim.Adam(ref.parameters(), lr=LR)                           # train.py:160
    a = types.SimpleNamespace(mixed_precision=False, smoothing=False, clip=False)
    idx2ans = {i: "a" for i in range(C)}
    train_loss, PREDS, acc, bleu, _ = VU.train_one_epoch(loader, ref, opt, crit, "cpu", None, a, idx2ans)
    extra = dict(seed=61, wsum=ws, dims=[B, T, hw, C], lr=LR, losses=np.array(crit.values), mean_loss=train_loss,
                 preds=PREDS, total_acc=acc)
    for i, b in enumerate(batches):
        for n, t in zip(("img", "ids", "seg", "mask", "tgt"), b):
            extra[f"{n}{i}"] = t
    finish("loop_vqa", ref, extra, ["transformer.mains.0.proj.weight", "transformer.mains.1.ff.0.weight"])


# answers in pairs (2k, 2k+1) that share words (partial BLEU-1 credit, brevity penalty, clipping) or share none
EVAL_ANSWERS = ["ct with contrast", "ct", "yes", "no", "mr flair", "mr t2 weighted flair", "left lung", "right lung",
                "axial", "coronal", "pulmonary embolism", "embolism", "us doppler", "xr plain film", "brain", "heart",
                "t1", "pe", "sagittal", "angiogram", "the", "an the the", "lung"]
EVAL_CATEGORIES = ["binary", "plane", "modality", "abnormality", "plane", "modality"]   # no 'organ': nan in the reference


def loop_vqa_eval():
    """vqamed2019/utils.py:690-767 (validate) and :769-843 (test) run as they stand: eval-mode forward (BatchNorm on its
    running buffers, which are moved off their defaults first), mean of the per-batch losses, softmax(1).argmax(1),
    total + per-category accuracy / BLEU-1 rounded to 4, and the --category form (one number each)."""
    import pandas as pd
    C, B, T, hw, nb = 23, 4, 10, 64, 3
    kw = dict(transformer_model="realformer", dataset="VQA-Med", hidden_size=768, n_layers=2, heads=12,
              hidden_dropout_prob=0.0, vocab_size=C, resnet_layers=(1, 1, 1, 1), resnet_width=64, bert_max_pos=32,
              use_relu=False, cnn_encoder="resnet152")
    args = O.make_args(**kw)
    torch.manual_seed(61)
    orc = O.OracleModel(args)
    LO.perturb_bn_buffers(orc, seed=62)
    ref = MG.build_ref_model(args, orc)
    MG.zero_dropout(ref)
    batches = [synth.vqa_batch(B, T, hw, vocab=C, n_classes=C, seed=95 + i) for i in range(nb)]
    # targets: what the model predicts, except four samples that get the paired answer (so that every kind of accuracy
    # and BLEU value occurs: 100, 50, 75, partial unigram credit)
    ref.eval()
    fixed = []
    with torch.no_grad():
        for bi, (img, ids, seg, mask, _) in enumerate(batches):
            pred = ref(img, ids, seg, mask)[0].softmax(1).argmax(1)
            j = torch.arange(B) + bi * B
            miss = (j == 1) | (j == 4) | (j == 6) | (j == 11)
            tgt = torch.where(miss, torch.where(pred == C - 1, pred - 1, pred ^ 1), pred)
            fixed.append((img, ids, seg, mask, tgt))
    batches = fixed
    cats = np.array([EVAL_CATEGORIES[i % len(EVAL_CATEGORIES)] for i in range(B * nb)])
    val_df = pd.DataFrame({"category": cats})
    idx2ans = {i: a for i, a in enumerate(EVAL_ANSWERS)}
    assert len(idx2ans) == C
    VU.sentence_bleu = lambda refs, hyp, weights=None: LO.sentence_bleu_unigram(refs, hyp)
    loader = [(img, ids.unsqueeze(1), seg, mask.unsqueeze(1), tgt, ["p"] * B) for img, ids, seg, mask, tgt in batches]
    a = types.SimpleNamespace(mixed_precision=False, smoothing=False, category=None)
    extra = dict(seed=61, bn_seed=62, dims=[B, T, hw, C, nb], categories=cats, answers=np.array(EVAL_ANSWERS))
    for name, fn, crit in (("val", VU.validate, Recorder(ASLSingleLabel())),          # train.py:172-174 / :230
                           ("test", VU.test, Recorder(torch.nn.CrossEntropyLoss()))):  # eval.py:128,144
        loss, PREDS, acc, bleu = fn(loader, ref, crit, "cpu", None, a, val_df, idx2ans)
        extra.update({f"{name}_loss": loss, f"{name}_preds": PREDS, f"{name}_batch_losses": np.array(crit.values),
                      f"{name}_acc_keys": np.array(list(acc.keys())), f"{name}_acc_vals": np.array(list(acc.values()), dtype=np.float64),
                      f"{name}_bleu_keys": np.array(list(bleu.keys())), f"{name}_bleu_vals": np.array(list(bleu.values()), dtype=np.float64)})
        assert not ref.training
    # --category <name> (utils.py:741-743 / :813-815): plain numbers
    a.category = "plane"
    loss, PREDS, acc, bleu = VU.validate(loader, ref, Recorder(ASLSingleLabel()), "cpu", None, a, val_df, idx2ans)
    extra.update(cat_acc=acc, cat_bleu=bleu)
    for i, b in enumerate(batches):
        for n, t in zip(("img", "ids", "seg", "mask", "tgt"), b):
            extra[f"{n}{i}"] = t
    MG.save("loop_vqa_eval", **extra)


if __name__ == "__main__":
    loop_mlm()
    loop_supcon()
    loop_vqa()
    loop_vqa_eval()
