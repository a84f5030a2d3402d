"""Evaluation path of the VQA-Med-2019 loop: validate / test with per-category accuracy and
BLEU-1, as vqamed2019/utils.py:690-843 computes them (callers: vqamed2019/train.py:230-231, eval.py:107-178).

  sentence_bleu1        nltk.translate.bleu_score.sentence_bleu([ref], hyp, weights=[1]) as utils.py:328-330 calls it
                        (nltk is a third-party dependency, unpinned by the reference and absent from the build image:
                        restated from its published algorithm -- clipped unigram precision x brevity penalty, 0 when no
                        unigram matches)
  calculate_bleu_score  utils.py:328-330 (mean BLEU-1 of idx2ans[pred] against idx2ans[target])
  category_metrics      utils.py:740-765 / 813-841: total + binary / plane / organ / modality / abnormality, rounded to 4
  validate, test        utils.py:690-767, 769-843: eval-mode forward, mean of the per-batch losses, softmax(1).argmax(1)
  write_test_files      eval.py:171-178: <model>_preds.csv (the test table + preds / decode_preds / decode_ans) and
                        <model>_res.txt (image stem | decoded prediction), byte-compatible with what pandas writes there

The forward pass is the HIP engine (mmvqa_amd.Model); the metrics are host bookkeeping on the predicted class ids.
"""
from __future__ import annotations

import csv
import math
import os
import warnings
from collections import Counter

import numpy as np
import torch

CATEGORIES = ("binary", "plane", "organ", "modality", "abnormality")
_SHORT = {"abnormality": "abnorm"}


def sentence_bleu1(reference_tokens, hypothesis_tokens):
    """BLEU with weights [1] and one reference, no smoothing (nltk's default SmoothingFunction.method0)"""
    hyp_len, ref_len = len(hypothesis_tokens), len(reference_tokens)
    counts = Counter(hypothesis_tokens)
    ref_counts = Counter(reference_tokens)
    num = sum(min(c, ref_counts[w]) for w, c in counts.items())      # clipped unigram matches
    den = max(1, hyp_len)
    if num == 0:
        return 0.0
    if hyp_len > ref_len:
        bp = 1.0
    elif hyp_len == 0:
        bp = 0.0
    else:
        bp = math.exp(1.0 - ref_len / hyp_len)
    return bp * math.exp(math.fsum([1.0 * math.log(num / den)]))


def calculate_bleu_score(preds, targets, idx2ans):
    """utils.py:328-330; np.mean of an empty selection is nan, as in the reference"""
    per = np.asarray([sentence_bleu1(idx2ans[int(t)].split(), idx2ans[int(p)].split()) for p, t in zip(preds, targets)])
    return np.mean(per)


def category_metrics(preds, targets, categories, idx2ans, prefix=""):
    """(acc dict, bleu dict) with the reference's key names: prefix 'val_' in validate (utils.py:747-765), '' in test
    (:821-841)"""
    preds, targets = np.asarray(preds), np.asarray(targets)
    cats = np.asarray(categories)
    acc = {prefix + "total_acc": np.round((preds == targets).mean() * 100., 4)}
    bleu = {prefix + "total_bleu": np.round(calculate_bleu_score(preds, targets, idx2ans), 4)}
    for c in CATEGORIES:
        sel = cats == c
        short = _SHORT.get(c, c)
        with np.errstate(invalid="ignore"), warnings.catch_warnings():
            warnings.simplefilter("ignore")          # an empty category gives nan ("Mean of empty slice"), as upstream
            acc[prefix + short + "_acc"] = np.round((preds[sel] == targets[sel]).mean() * 100., 4)
            bleu[prefix + short + "_bleu"] = np.round(calculate_bleu_score(preds[sel], targets[sel], idx2ans), 4)
    return acc, bleu


@torch.no_grad()
def _run(loader, model, criterion, categories, idx2ans, category, prefix):
    model.eval()
    losses, PREDS, TARGETS = [], [], []
    for img, question_token, segment_ids, attention_mask, target in loader:
        logits, _, _ = model(img, question_token, segment_ids, attention_mask)       # utils.py:711 / 789
        loss = criterion(logits, target)
        losses.append(loss.detach().cpu().numpy())
        PREDS.append(logits.softmax(1).argmax(1).detach())                            # utils.py:721 / 800
        TARGETS.append(target)
    loss = np.mean(losses)
    P, T = torch.cat(PREDS).cpu().numpy(), torch.cat(TARGETS).cpu().numpy()
    if category:                                                                      # --category <name>: one number each
        return loss, P, (P == T).mean() * 100., calculate_bleu_score(P, T, idx2ans)
    acc, bleu = category_metrics(P, T, categories, idx2ans, prefix)
    return loss, P, acc, bleu


def validate(loader, model, criterion, categories, idx2ans, category=None):
    """utils.py:690-767 -> (val_loss, PREDS, acc, bleu); `categories` = val_df['category'] in loader order"""
    return _run(loader, model, criterion, categories, idx2ans, category, "val_")


def test(loader, model, criterion, categories, idx2ans, category=None):
    """utils.py:769-843 -> (test_loss, PREDS, acc, bleu)"""
    return _run(loader, model, criterion, categories, idx2ans, category, "")


def write_test_files(rows, columns, predictions, idx2ans, out_dir, model_name):
    """The two files of the reference's test-set run (vqamed2019/eval.py:171-178).
    rows: the test split as a list of records in loader order, `columns` their field names (img_id, question, answer =
    class id, category, mode as load_data() leaves them, utils.py:51-79).  Files (same names, columns and separators):
      <out_dir>/<model_name>_preds.csv  header + one line per sample: the table's own columns, then preds (class id),
                                        decode_preds, decode_ans; comma separated, minimal quoting, '\n' line ends
      <out_dir>/<model_name>_res.txt    no header: <image file stem>|<decoded prediction>
    Returns the two paths."""
    os.makedirs(out_dir, exist_ok=True)
    ia, ii = columns.index("answer"), columns.index("img_id")
    p_csv = os.path.join(out_dir, f"{model_name}_preds.csv")
    p_res = os.path.join(out_dir, f"{model_name}_res.txt")
    with open(p_csv, "w", newline="") as f:
        w = csv.writer(f, lineterminator="\n")
        w.writerow(list(columns) + ["preds", "decode_preds", "decode_ans"])
        for rec, p in zip(rows, predictions):
            w.writerow(list(rec) + [int(p), idx2ans[int(p)], idx2ans[int(rec[ia])]])
    with open(p_res, "w", newline="") as f:
        w = csv.writer(f, delimiter="|", lineterminator="\n")
        for rec, p in zip(rows, predictions):
            stem = str(rec[ii]).split("/")[-1].split(".")[0]
            w.writerow([stem, idx2ans[int(p)]])
    return p_csv, p_res
