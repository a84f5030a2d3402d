"""CPU restatement of the reference's three training-step loops (the CALLERS of the hot path).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference modules that hold these loops cannot be imported in the build image (top-level imports of wandb /
nltk / pytorch_lightning / sentence_transformers / googletrans / bert_score), so they are restated
from the source text; every function cites the lines it follows.  Pinning: the arithmetic inside a step (model
forward, the three losses, Adam) IS pinned by reference-generated fixtures (tests/golden/, oracle/mmbert_oracle.py);
what this file adds is the ORDER of operations of a step and the bookkeeping around it, which has no fixture
upstream (the reference has no tests) => "parity unpinned" for the loop order itself, restated line by line.

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
