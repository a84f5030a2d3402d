"""Synthetic inputs with the layout of the reference's data layer.

ROCO / MLM (pretrain/roco_utils.py:162-199): tokens [CLS] + 5 visual slots (id 0) + [SEP] + caption +
[SEP] + padding; segment ids 0*7, 1*(n+1), 0*pad; mask 1*(8+n); labels = original id at [MASK]ed caption
positions, 0 elsewhere.  VQA-Med (vqamed2019/utils.py:156-170): same token layout, one class id per sample.
"""
from __future__ import annotations

import torch


def roco_batch(B, T=32, hw=224, vocab=30522, seed=1234, device="cpu", mlm_prob=0.15):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 3, hw, hw, generator=g) * 2 - 1
    ids = torch.zeros(B, T, dtype=torch.long)
    seg = torch.zeros(B, T, dtype=torch.long)
    mask = torch.zeros(B, T, dtype=torch.long)
    tgt = torch.zeros(B, T, dtype=torch.long)
    lo = min(1000, max(1, vocab // 2))
    for b in range(B):
        n_max = T - 8                      # [CLS] + 5 visual slots + 2x[SEP] are always present
        n = int(torch.randint(min(4, n_max), n_max + 1, (1,), generator=g))
        cap = torch.randint(lo, vocab, (n,), generator=g)
        m = torch.rand(n, generator=g) < mlm_prob
        ids[b, 0] = 101 % vocab
        ids[b, 6] = 102 % vocab
        ids[b, 7:7 + n] = torch.where(m, torch.full_like(cap, 103 % vocab), cap)
        ids[b, 7 + n] = 102 % vocab
        tgt[b, 7:7 + n] = torch.where(m, cap, torch.zeros_like(cap))
        seg[b, 7:8 + n] = 1
        mask[b, :8 + n] = 1
    return tuple(t.to(device) for t in (img, ids, seg, mask, tgt))


def vqa_batch(B, T=32, hw=224, vocab=30522, n_classes=1552, seed=1234, device="cpu"):
    img, ids, seg, mask, _ = roco_batch(B, T, hw, vocab, seed, "cpu", mlm_prob=0.0)
    g = torch.Generator().manual_seed(seed + 77)
    tgt = torch.randint(0, n_classes, (B,), generator=g)
    return tuple(t.to(device) for t in (img, ids, seg, mask, tgt))


VQA_CATEGORIES = ("modality", "plane", "organ", "abnormality", "binary")   # the five question types of VQA-Med-2019


def vqa_test_table(n, n_classes, seed=1234, data_dir="../ImageClef-2019-VQA-Med"):
    """A synthetic stand-in for the test split's table as vqamed2019/utils.py:51-79 (load_data) + eval.py:84-97 leave
    it: columns (img_id, question, answer, category, mode) with img_id a path under <data_dir>/Test/images, answer the
    class id.  Returns (columns, rows, idx2ans)."""
    g = torch.Generator().manual_seed(seed + 4242)
    ans = torch.randint(0, n_classes, (n,), generator=g).tolist()
    cat = torch.randint(0, len(VQA_CATEGORIES), (n,), generator=g).tolist()
    cols = ["img_id", "question", "answer", "category", "mode"]
    rows = [(f"{data_dir}/Test/images/synpic{10000 + i}.jpg", f"what is shown in image {i}?", ans[i], VQA_CATEGORIES[cat[i]], "test")
            for i in range(n)]
    idx2ans = {i: f"answer {i}" if i % 3 else f"finding, type {i}" for i in range(n_classes)}
    return cols, rows, idx2ans
