"""Checkpoint compatibility: local files in the formats the reference's loaders fetch or
write, into the flat-buffer Model -- no network.

  load_roco_pretrained   vqamed2019/train.py:125-135  (key-filtered load of a ROCO-pretrained Model state_dict; the
                         caller then swaps classifier[2], :137)
  load_model             vqamed2019/train.py:139-144, vqamed2019/eval.py:109-112 (strict load of a Model state_dict)
  load_backbone          what `models.resnet152(pretrained=True)` / `timm.create_model('tf_efficientnetv2_m',
                         features_only=True, pretrained=True)` put into `transformer.trans.model`
                         (models/image_encoding.py:20-26): a torchvision ResNet-152 state_dict, or a timm
                         tf_efficientnetv2_m state_dict (full classifier model or features-only; conv_head / bn2 /
                         classifier are dropped as features_only does)
  load_bert_embeddings   what `AutoModel.from_pretrained('bert-base-uncased')` contributes (models/mmbert.py:52-56:
                         only children()[0] = BertEmbeddings is kept): word / position / token_type tables + LayerNorm
  read_state_dict        .pt / .pth / .bin (torch.save) or .safetensors

Every loader checks shapes and reports what it did; parameters are copied INTO the model's flat buffers (the
nn.Parameter views keep pointing at them).
"""
from __future__ import annotations

import torch


def read_state_dict(path_or_dict):
    if isinstance(path_or_dict, dict):
        sd = path_or_dict
    elif str(path_or_dict).endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(str(path_or_dict))
    else:
        sd = torch.load(str(path_or_dict), map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "model" in sd and "optimizer" in sd and isinstance(sd["model"], dict):
        sd = sd["model"]                  # a "recorder" dict (pretrain/roco_train.py:165-171)
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    return sd


def load_roco_pretrained(model, path_or_dict):
    """train.py:125-135: keep the checkpoint's entries whose KEY exists in the model, overwrite, load.  As in the
    reference a kept entry with a different shape is an error (load_state_dict raises).  Returns the sorted lists
    (loaded, skipped_checkpoint_keys, untouched_model_keys)."""
    model_dict = model.state_dict()
    pretrained = read_state_dict(path_or_dict)
    kept = {k: v for k, v in pretrained.items() if k in model_dict}
    skipped = sorted(k for k in pretrained if k not in model_dict)
    untouched = sorted(k for k in model_dict if k not in kept)
    model_dict.update(kept)
    model.load_state_dict(model_dict)
    return sorted(kept), skipped, untouched


def load_model(model, path_or_dict):
    """strict load of a full Model state_dict (train.py:144, eval.py:112)"""
    model.load_state_dict(read_state_dict(path_or_dict), strict=True)


_TIMM_HEAD = ("conv_head.", "bn2.", "classifier.")      # dropped by features_only=True
_TV_PREFIXES = ("module.", "model.")


def _strip(sd, prefixes):
    out = {}
    for k, v in sd.items():
        for p in prefixes:
            if k.startswith(p):
                k = k[len(p):]
        out[k] = v
    return out


def load_backbone(model, path_or_dict):
    """torchvision resnet152 / timm tf_efficientnetv2_m weights -> transformer.trans.model.* (including the BatchNorm
    buffers and, for ResNet, the unused fc.*).  Raises if a body tensor is missing or has another shape."""
    sd = _strip(read_state_dict(path_or_dict), _TV_PREFIXES)
    sd = {k: v for k, v in sd.items() if not k.startswith(_TIMM_HEAD)}
    prefix = "transformer.trans.model."
    own = {k[len(prefix):]: v for k, v in model.state_dict().items() if k.startswith(prefix)}
    missing = sorted(k for k in own if k not in sd)
    unexpected = sorted(k for k in sd if k not in own)
    if missing or unexpected:
        raise KeyError(f"backbone checkpoint does not match {type(model).__name__}'s backbone: missing {missing[:5]} "
                       f"({len(missing)}), unexpected {unexpected[:5]} ({len(unexpected)})")
    bad = [k for k in own if tuple(own[k].shape) != tuple(sd[k].shape)]
    if bad:
        raise ValueError(f"backbone checkpoint shapes differ: {[(k, tuple(sd[k].shape), tuple(own[k].shape)) for k in bad[:5]]}")
    model.load_state_dict({prefix + k: v for k, v in sd.items()}, strict=False)
    return len(sd)


_EMB_KEYS = ("word_embeddings.weight", "position_embeddings.weight", "token_type_embeddings.weight",
             "LayerNorm.weight", "LayerNorm.bias")


def load_bert_embeddings(model, path_or_dict):
    """HF BertModel / BertForMaskedLM state_dict -> transformer.bert_embedding.* (the encoder layers and the pooler
    of the checkpoint are not used by the reference, mmbert.py:55-56).  Old checkpoints name the LayerNorm
    parameters gamma / beta."""
    sd = read_state_dict(path_or_dict)
    found = {}
    for k, v in sd.items():
        kk = k.replace("LayerNorm.gamma", "LayerNorm.weight").replace("LayerNorm.beta", "LayerNorm.bias")
        for e in _EMB_KEYS:
            if kk.endswith("embeddings." + e):
                found[e] = v
    missing = [e for e in _EMB_KEYS if e not in found]
    if missing:
        raise KeyError(f"no BERT embedding tensors {missing} in the checkpoint")
    prefix = "transformer.bert_embedding."
    own = model.state_dict()
    for e, v in found.items():
        if tuple(own[prefix + e].shape) != tuple(v.shape):
            raise ValueError(f"{e}: checkpoint {tuple(v.shape)} vs model {tuple(own[prefix + e].shape)}")
    model.load_state_dict({prefix + e: v for e, v in found.items()}, strict=False)
    return sorted(found)
