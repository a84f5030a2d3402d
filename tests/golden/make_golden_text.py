#!/usr/bin/env python3
"""Golden vectors for the text side and the step loops (8(a) row a20), produced by the
REFERENCE's own functions.  Run ONCE in the build container:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_text.py

pretrain/roco_utils.py and vqamed2019/utils.py import packages the image lacks (wandb, nltk, pytorch_lightning,
torchvision, matplotlib is present); those are registered as NAME-ONLY stubs (no arithmetic) after `transformers`
has been imported, exactly like torchvision/timm in make_golden.py.  The tokenizer is HF's BertTokenizer
(`tokenizers` backend: BertNormalizer + BertPreTokenizer + WordPiece) on the small synthetic vocabulary
tests/golden/text_vocab.txt (no network: the real bert-base-uncased vocab.txt is not in the image).

Writes tests/golden/text.json: inputs + the reference's outputs (token strings, ids, labels, masks).  Data only.
"""
import json
import os
import pickle
import random
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from transformers import BertTokenizer, BertModel, AutoTokenizer, AutoModel  # noqa: E402,F401  (before the stubs)
import torch.nn as nn  # noqa: E402


def stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def register_stubs():
    stub("wandb")
    nltk, tr = stub("nltk"), stub("nltk.translate")
    bs = stub("nltk.translate.bleu_score", sentence_bleu=lambda *a, **k: None)   # never called by this script
    nltk.translate, tr.bleu_score = tr, bs

    class _LM(nn.Module):
        pass

    stub("pytorch_lightning", LightningModule=_LM, LightningDataModule=object, Trainer=object)
    tv, tvt, tvm = stub("torchvision"), stub("torchvision.transforms"), stub("torchvision.models")
    tv.transforms, tv.models = tvt, tvm
    tvm.resnet152 = lambda **k: None
    stub("timm", create_model=lambda *a, **k: None)


def make_vocab():
    """deterministic synthetic vocabulary: specials, punctuation, letters/digits and their ## forms, word pieces"""
    toks = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    toks += list("!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~")
    toks += list("0123456789") + list("abcdefghijklmnopqrstuvwxyz")
    toks += ["##" + c for c in "0123456789abcdefghijklmnopqrstuvwxyz"]
    words = """the a of in with and no is are on at to from showing shows show seen noted image images ct mri scan x ray
    chest abdomen pelvis brain left right upper lower lobe lung lungs heart liver kidney spleen bone fracture mass lesion
    nodule opacity effusion pleural pneumonia pneumo thorax edema tumor cyst normal abnormal axial coronal sagittal
    contrast enhanced patient year old male female arrow arrows mm cm large small bilateral cafe naive resume
    what which where how organ plane modality abnormality does this yes""".split()
    toks += words
    toks += ["##s", "##ing", "##ed", "##nia", "##al", "##ly", "##tion", "##ic", "##ous", "##er", "##est", "##thorax",
             "##gram", "##graphy", "##scan", "##mm", "##cm", "##oma", "##itis", "##osis"]
    seen, out = set(), []
    for t in toks:
        if t not in seen:
            seen.add(t)
            out.append(t)
    return out


SENTENCES = [
    "The chest X-ray shows pneumonia in the left lower lobe.",
    "Axial CT scan of the abdomen showing a large liver mass (arrow).",
    "No pleural effusion or pneumothorax is seen.",
    "MRI brain: 12mm lesion, contrast-enhanced, coronal plane",
    "café naïve résumé Ångström patient",
    "Bilateral   opacities\tnoted\n in lungs",
    "xyzzyq unknownword 中文 mixed   nbsp and​zero width",
    "tumor. mass. cyst, nodule; fracture: edema!",
    "a " + "b" * 120 + " c",
    "heart heart heart liver liver kidney spleen bone bone lesion lesion lesion nodule opacity effusion tumor cyst mass",
    "",
    "What is the modality of this image?",
    "which organ is abnormal in this ct scan of a 45 year old male patient with a very long question text that keeps going on and on",
    "[MASK] is written out, and so is [SEP] inside text",
    "##nia ##s # hash",
]
MED_VOCAB = {"organ": ["heart", "liver", "kidney", "spleen", "lung", "lungs", "bone"],
             "finding": ["pneumonia", "mass", "lesion", "nodule", "opacity", "effusion", "fracture", "edema", "tumor", "cyst",
                         "pneumothorax"],
             "modality": ["ct", "mri", "x-ray", "CT", "MRI"]}


def main():
    vocab = make_vocab()
    vpath = os.path.join(HERE, "text_vocab.txt")
    with open(vpath, "w", encoding="utf-8") as f:
        f.write("\n".join(vocab) + "\n")
    tok = BertTokenizer(vocab={t: i for i, t in enumerate(vocab)})
    register_stubs()
    sys.path.insert(0, "/root/reference")
    sys.path.insert(0, "/root/reference/pretrain")
    sys.path.insert(0, "/root/reference/vqamed2019")
    import importlib
    RU = importlib.import_module("pretrain.roco_utils")
    VU = importlib.import_module("vqamed2019.utils")

    out = {"sentences": SENTENCES, "med_vocab": MED_VOCAB}
    # tokenizer surface the reference uses: tokenize / encode / convert_tokens_to_ids
    out["tokenize"] = [tok.tokenize(s) for s in SENTENCES]
    out["encode"] = [tok.encode(s) for s in SENTENCES]
    out["special_ids"] = [tok.cls_token_id, tok.sep_token_id, tok.mask_token_id, tok.pad_token_id, tok.unk_token_id]
    # random unicode strings: the normaliser / pre-tokeniser corner cases
    rng = random.Random(7)
    pool = ("abc XYZ 019 .,;:!?()[]{}-_/\\'\"#@ \t\n" + "éèüñçÅØßłİı" +
            "́̈  ​‍﻿­—’“、中文日本АбΩω�\x00\x07")
    fuzz = ["".join(rng.choice(pool) for _ in range(rng.randint(1, 40))) for _ in range(300)]
    out["fuzz"] = fuzz
    out["fuzz_tokens"] = [tok.tokenize(s) for s in fuzz]
    # get_keywords (roco_utils.py:47-63) reads data_dir/vocab/med_vocab.pkl
    d = tempfile.mkdtemp()
    os.makedirs(os.path.join(d, "vocab"))
    with open(os.path.join(d, "vocab", "med_vocab.pkl"), "wb") as f:
        pickle.dump(MED_VOCAB, f)
    args = types.SimpleNamespace(data_dir=d, mlm_prob=0.5, num_vis=5, max_position_embeddings=32, task="MLM")
    keywords = RU.get_keywords(args)
    out["keywords_sorted"] = sorted(keywords)
    # mask_word / encode_text with a fixed `random` stream (the reference draws from the module-level generator)
    cases = []
    for seed, prob, T in ((1, 0.5, 32), (2, 0.15, 32), (3, 1.0, 20), (4, 0.0, 75), (5, 0.5, 12)):
        args.mlm_prob, args.max_position_embeddings = prob, T
        random.seed(seed)
        mw = [RU.mask_word(s, tok, keywords, args) for s in SENTENCES]
        random.seed(seed)
        et = [RU.encode_text(s, tok, keywords, args, None) for s in SENTENCES]
        cases.append(dict(seed=seed, mlm_prob=prob, T=T, mask_word=[[a, b] for a, b in mw],
                          encode_text=[[t.tolist() for t in e] for e in et]))
    out["mlm_cases"] = cases
    # VQA encode_text (vqamed2019/utils.py:156-170)
    vq = []
    for T in (28, 12, 32):
        a = types.SimpleNamespace(max_position_embeddings=T)
        vq.append(dict(T=T, rows=[list(VU.encode_text(s, tok, a)) for s in SENTENCES]))
    out["vqa_cases"] = vq
    with open(os.path.join(HERE, "text.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=True, indent=0)
    print("wrote text.json:", len(SENTENCES), "sentences,", len(fuzz), "fuzz strings,", len(vocab), "vocab entries")


if __name__ == "__main__":
    main()
