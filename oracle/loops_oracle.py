"""CPU restatement of the reference's three training-step loops (the CALLERS of the hot path).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Every function cites the reference lines it follows.  PINNED: tests/golden/make_golden_loops.py imports the
reference's own pretrain/roco_utils.py, models/SupConLoss/supcon_utils.py and vqamed2019/utils.py (packages the image
lacks -- wandb, nltk, pytorch_lightning, sentence_transformers, googletrans, bert_score, torchvision, timm -- as
name-only stubs), runs THEIR train_one_epoch functions on the reference Model for two batches and records per-step
losses, accuracy and the parameters after the two Adam steps (tests/golden/loop_*.npz);
tests/test_oracle_golden.py::test_loops_match_reference_loops replays them through this file.
The evaluation half (validate / test, vqamed2019/utils.py:690-843) is pinned the same way (loop_vqa_eval.npz: the
reference's own functions on a loader with mixed categories, one of them empty); only nltk's sentence_bleu inside it
is absent from the image: BLEU-1 is restated from the published algorithm, marked "parity unpinned" below and anchored
by hand-computed known answers; the fixture script hands this restatement to the reference's calculate_bleu_score.

A "loader" here is any iterable of batches with the reference's tuple layout (already on the CPU).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from oracle import mmbert_oracle as O


def mlm_train_one_epoch(loader, model, criterion, optimizer):
    """pretrain/roco_utils.py:207-290 with args.task == 'MLM', args.mixed_precision False.
    criterion = nn.NLLLoss() (pretrain/roco_train.py:89).  Returns (mean loss, total accuracy %, per-step losses,
    per-step predictions at target > 0)."""
    model.train()                                                       # :209
    train_loss, PREDS, TARGETS = [], [], []
    for img, caption_token, segment_ids, attention_mask, target in loader:   # :214
        optimizer.zero_grad()                                           # :222
        logits = model(img, caption_token, segment_ids, attention_mask)  # :233
        logits = logits.log_softmax(-1)                                 # :235
        loss = criterion(logits.permute(0, 2, 1), target)               # :236
        loss.backward()                                                 # :246
        optimizer.step()                                                # :247
        bool_label = target > 0                                         # :257
        pred = logits[bool_label, :].argmax(1)                          # :259
        valid_labels = target[bool_label]                               # :260
        PREDS.append(pred)
        TARGETS.append(valid_labels)
        train_loss.append(loss.detach().cpu().numpy())                  # :267
    P, T = torch.cat(PREDS).cpu().numpy(), torch.cat(TARGETS).cpu().numpy()   # :281-282
    total_acc = (P == T).mean() * 100. if len(P) else float("nan")      # :285
    return np.mean(train_loss), total_acc, train_loss, PREDS


def process_tensors(img, caption_token, aug_tokens, segment_ids, attention_mask, target, aug_targets):
    """models/SupConLoss/supcon_utils.py:253-256"""
    def cat_tensors(a, b):
        return torch.cat([a, b], dim=0)
    return (cat_tensors(img[0], img[1]), cat_tensors(caption_token, aug_tokens), cat_tensors(segment_ids, segment_ids),
            cat_tensors(attention_mask, attention_mask), cat_tensors(target, aug_targets))


def supcon_train_one_epoch(loader, model, criterion, supcon_loss, optimizer):
    """models/SupConLoss/supcon_utils.py:263-323.  The similarity mask of :286 is built and then NOT passed
    (:287 calls supcon_loss(feat)) => SimCLR form; it is therefore not restated.  supcon_loss = O.supcon_simclr."""
    model.train()
    train_loss, PREDS, TARGETS = [], [], []
    for img, caption_token, aug_tokens, segment_ids, attention_mask, target, aug_targets in loader:   # :270
        img, caption_token, segment_ids, attention_mask, target = process_tensors(
            img, caption_token, aug_tokens, segment_ids, attention_mask, target, aug_targets)        # :271
        optimizer.zero_grad()                                           # :278
        logits, feat = model(img, caption_token, segment_ids, attention_mask)   # :280
        logits = logits.log_softmax(-1)                                 # :281
        loss = criterion(logits.permute(0, 2, 1), target)               # :282
        bsz = img.shape[0] // 2                                         # :284
        feat = O.split_feat(feat, bsz)                                  # :285
        loss_supcon = supcon_loss(feat)                                 # :287
        loss = loss + loss_supcon                                       # :289
        loss.backward()                                                 # :292
        optimizer.step()                                                # :294
        bool_label = target > 0                                         # :297
        if bool_label.any():
            pred = logits[bool_label, :].argmax(1)
            PREDS.append(pred)
            TARGETS.append(target[bool_label])
        train_loss.append(loss.detach().cpu().numpy())
    P, T = torch.cat(PREDS).cpu().numpy(), torch.cat(TARGETS).cpu().numpy()
    return np.mean(train_loss), (P == T).mean() * 100., train_loss, PREDS


def vqa_train_one_epoch(loader, model, optimizer, criterion, clip=False):
    """vqamed2019/utils.py:625-688 with args.mixed_precision False, args.smoothing False.
    criterion = nn.CrossEntropyLoss() | ASLSingleLabel (vqamed2019/train.py:164-174)."""
    model.train()
    train_loss, PREDS, TARGETS = [], [], []
    for img, question_token, segment_ids, attention_mask, target in loader:     # :633 (imgid, category unused here)
        optimizer.zero_grad()                                           # :639
        logits, _, _ = model(img, question_token, segment_ids, attention_mask)  # :646
        loss = criterion(logits, target)                                # :650
        loss.backward()                                                 # :661
        if clip:
            nn.utils.clip_grad_norm_(model.parameters(), 1.0)           # :663-664
        optimizer.step()                                                # :666
        TARGETS.append(target)
        PREDS.append(logits.softmax(1).argmax(1).detach())              # :673
        train_loss.append(loss.detach().cpu().numpy())
    P, T = torch.cat(PREDS).cpu().numpy(), torch.cat(TARGETS).cpu().numpy()
    return np.mean(train_loss), (P == T).mean() * 100., train_loss, PREDS


def perturb_bn_buffers(model, seed):
    """test helper: eval-mode BatchNorm reads its running buffers; a freshly built model has them at (0, 1), where a
    mistake in their use is invisible.  Moves every BatchNorm2d buffer to seeded values (the fixture script and the
    replaying tests call this on the same seeded oracle model)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)


# --------------------------------------------------------------------------- evaluation (vqamed2019/utils.py:690-843)
def sentence_bleu_unigram(references, hypothesis):
    """nltk.translate.bleu_score.sentence_bleu(references, hypothesis, weights=[1]) -- third-party (nltk, version
    unpinned by the reference, absent from the build image): restated from the published algorithm (Papineni et al.
    2002 as nltk implements it): modified unigram precision with clipping against the references' max counts,
    closest reference length, brevity penalty exp(1 - r/c) for c <= r, result 0 when no unigram matches.
    "parity unpinned" (no nltk here): anchored by hand-computed known answers in tests/test_evaluate.py."""
    import math
    from collections import Counter
    counts = Counter(hypothesis)
    max_counts = {}
    for ref in references:
        rc = Counter(ref)
        for w in counts:
            max_counts[w] = max(max_counts.get(w, 0), rc[w])
    numerator = sum(min(c, max_counts[w]) for w, c in counts.items())
    denominator = max(1, sum(counts.values()))
    hyp_len = len(hypothesis)
    ref_len = min((len(r) for r in references), key=lambda rl: (abs(rl - hyp_len), rl))
    if numerator == 0:
        return 0
    if hyp_len > ref_len:
        bp = 1
    elif hyp_len == 0:
        bp = 0
    else:
        bp = math.exp(1 - ref_len / hyp_len)
    return bp * math.exp(math.fsum([1 * math.log(numerator / denominator)]))


def calculate_bleu_score(preds, targets, idx2ans):
    """vqamed2019/utils.py:328-330"""
    bleu_per_answer = np.asarray([sentence_bleu_unigram([idx2ans[target].split()], idx2ans[pred].split())
                                  for pred, target in zip(preds, targets)])
    return np.mean(bleu_per_answer)


def vqa_validate(loader, model, criterion, val_category, idx2ans, prefix="val_", category=None):
    """vqamed2019/utils.py:690-767 (validate, prefix 'val_') and :769-843 (test, prefix ''),
    args.mixed_precision / args.smoothing False.  val_category = val_df['category'] as a numpy array of strings;
    category = args.category (set: one accuracy and one BLEU number, :741-743).
    PINNED by tests/golden/loop_vqa_eval.npz: the reference's own validate / test on a mixed-category loader."""
    model.eval()
    val_loss, PREDS, TARGETS = [], [], []
    with torch.no_grad():
        for img, question_token, segment_ids, attention_mask, target in loader:
            logits, _, _ = model(img, question_token, segment_ids, attention_mask)
            loss = criterion(logits, target)
            pred = logits.softmax(1).argmax(1).detach()
            PREDS.append(pred)
            TARGETS.append(target)
            val_loss.append(loss.detach().cpu().numpy())
        val_loss = np.mean(val_loss)
    PREDS = torch.cat(PREDS).cpu().numpy()
    TARGETS = torch.cat(TARGETS).cpu().numpy()
    if category:
        return val_loss, PREDS, (PREDS == TARGETS).mean() * 100., calculate_bleu_score(PREDS, TARGETS, idx2ans)
    cat = np.asarray(val_category)
    names = (("total", None), ("binary", "binary"), ("plane", "plane"), ("organ", "organ"), ("modality", "modality"),
             ("abnorm", "abnormality"))
    acc, bleu = {}, {}
    for short, c in names:
        sel = slice(None) if c is None else (cat == c)
        acc[prefix + short + "_acc"] = np.round((PREDS[sel] == TARGETS[sel]).mean() * 100., 4)
        bleu[prefix + short + "_bleu"] = np.round(calculate_bleu_score(PREDS[sel], TARGETS[sel], idx2ans), 4)
    return val_loss, PREDS, acc, bleu


def eval_write_csvs(test_df, predictions, idx2ans, out_dir, model_name):
    """vqamed2019/eval.py:171-178: the two files the test-set run leaves behind, written with pandas as the
    reference does.  test_df: the test split's DataFrame (columns img_id, question, answer (class id), category, mode)."""
    import os
    test_df = test_df.copy()
    test_df['preds'] = predictions                                                          # :171
    test_df['decode_preds'] = test_df['preds'].map(idx2ans)                                 # :172
    test_df['decode_ans'] = test_df['answer'].map(idx2ans)                                  # :173
    test_df.to_csv(os.path.join(out_dir, f'{model_name}_preds.csv'), index=False)           # :174
    result = test_df[['img_id', 'decode_preds']].copy()                                     # :176
    result['img_id'] = result['img_id'].apply(lambda x: x.split('/')[-1].split('.')[0])     # :177
    result.to_csv(os.path.join(out_dir, f'{model_name}_res.txt'), index=False, header=False, sep='|')   # :178
