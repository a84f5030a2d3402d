"""Checkpoint compatibility: files shaped like the ones the reference's loaders fetch
(torchvision resnet152, timm tf_efficientnetv2_m, HF bert-base-uncased) or write (Model state_dicts, recorder dicts)
load into the flat-buffer Model; the ROCO -> VQA-Med flow of vqamed2019/train.py:125-137 gives the same weights as the
reference's literal steps on the oracle model.  CPU only (parameter plumbing; the forward parity with loaded weights is
covered by the GPU tests, which all start from load_state_dict)."""
import os

import pytest
import torch

import mmvqa_amd
from mmvqa_amd import checkpoint as CK
from oracle import effnet_oracle as E
from oracle import mmbert_oracle as O


def small(**kw):
    d = dict(resnet_layers=(1, 2, 1, 1), resnet_width=8, hidden_size=96, n_layers=2, heads=12, vocab_size=50, emb_vocab=50,
             bert_max_pos=32)
    d.update(kw)
    return O.make_args(**d)


def randomize(m, seed):
    torch.manual_seed(seed)
    with torch.no_grad():
        for p in m.parameters():
            p.normal_(0, 0.1)
        for n, b in m.named_buffers():
            if b.dtype.is_floating_point:
                b.uniform_(0.5, 1.5)
            elif "num_batches_tracked" in n:
                b.fill_(7)
    return m


@pytest.mark.parametrize("ext", ["pt", "safetensors"])
def test_torchvision_resnet_file(tmp_path, ext):
    args = small()
    tv = randomize(O.OracleResNet(args.resnet_layers, args.resnet_width), 1)     # torchvision layout and key names
    path = tmp_path / ("resnet." + ext)
    if ext == "pt":
        torch.save(tv.state_dict(), path)
    else:
        from safetensors.torch import save_file
        save_file({k: v.contiguous() for k, v in tv.state_dict().items()}, str(path))
    m = mmvqa_amd.Model(args)
    before = m.state_dict()["fc1.weight"].clone()
    n = CK.load_backbone(m, str(path))
    sd = m.state_dict()
    assert n == len(tv.state_dict())
    for k, v in tv.state_dict().items():
        assert torch.equal(sd["transformer.trans.model." + k], v), k
    assert torch.equal(sd["fc1.weight"], before)                      # nothing else moved
    assert sd["transformer.trans.model.fc.weight"].shape == (1000, 8 * 32)   # torchvision's unused classifier travels too
    with pytest.raises(KeyError):
        CK.load_backbone(m, {k: v for k, v in tv.state_dict().items() if "layer3" not in k})
    with pytest.raises(ValueError):
        CK.load_backbone(m, {k: (v[:1] if k == "conv1.weight" else v) for k, v in tv.state_dict().items()})


def test_timm_effnet_file(tmp_path):
    args = small(cnn_encoder="tf_efficientnetv2_m", effnet_depth_div=8)
    body = randomize(E.OracleEffNetV2Features(8), 2)
    full = dict(body.state_dict())                                  # a classifier checkpoint as timm publishes it
    full.update({"conv_head.weight": torch.zeros(1280, 512, 1, 1), "bn2.weight": torch.ones(1280), "bn2.bias": torch.zeros(1280),
                 "bn2.running_mean": torch.zeros(1280), "bn2.running_var": torch.ones(1280),
                 "bn2.num_batches_tracked": torch.tensor(0), "classifier.weight": torch.zeros(1000, 1280),
                 "classifier.bias": torch.zeros(1000)})
    torch.save(full, tmp_path / "eff.pth")
    m = mmvqa_amd.Model(args)
    CK.load_backbone(m, str(tmp_path / "eff.pth"))
    sd = m.state_dict()
    for k, v in body.state_dict().items():
        assert torch.equal(sd["transformer.trans.model." + k], v), k


def test_hf_bert_file(tmp_path):
    from transformers import BertConfig, BertModel
    args = small(hidden_size=96, emb_vocab=50, vocab_size=50, bert_max_pos=32)
    torch.manual_seed(3)
    bert = BertModel(BertConfig(vocab_size=50, hidden_size=96, num_hidden_layers=1, num_attention_heads=4,
                                intermediate_size=64, max_position_embeddings=32))
    torch.save(bert.state_dict(), tmp_path / "pytorch_model.bin")
    m = mmvqa_amd.Model(args)
    got = CK.load_bert_embeddings(m, str(tmp_path / "pytorch_model.bin"))
    assert len(got) == 5
    sd, bsd = m.state_dict(), bert.state_dict()
    for e in CK._EMB_KEYS:
        assert torch.equal(sd["transformer.bert_embedding." + e], bsd["embeddings." + e]), e
    # BertForMaskedLM-style prefix and the old gamma/beta names
    old = {"bert." + k.replace("LayerNorm.weight", "LayerNorm.gamma").replace("LayerNorm.bias", "LayerNorm.beta"): v
           for k, v in bsd.items()}
    m2 = mmvqa_amd.Model(args)
    CK.load_bert_embeddings(m2, old)
    assert torch.equal(m2.state_dict()["transformer.bert_embedding.LayerNorm.weight"], bsd["embeddings.LayerNorm.weight"])
    with pytest.raises(ValueError):
        CK.load_bert_embeddings(mmvqa_amd.Model(small(emb_vocab=60)), str(tmp_path / "pytorch_model.bin"))


def test_roco_to_vqa_flow_equals_the_reference_steps(tmp_path):
    """vqamed2019/train.py:125-137 on a ROCO+SupCon-pretrained checkpoint: head.* is filtered out by key, everything
    else loads, then classifier[2] is replaced; same result as the literal steps on the oracle (reference-pinned) Model"""
    roco_args = small(transformer_model="realformer", supcon=True)
    pre = randomize(O.OracleModel(roco_args), 4)
    torch.save(pre.state_dict(), tmp_path / "roco.pt")
    vqa_args = small(transformer_model="realformer", dataset="VQA-Med")
    # the reference's literal steps on the oracle
    ref = O.OracleModel(vqa_args)
    model_dict = ref.state_dict()
    pretrained = {k: v for k, v in torch.load(tmp_path / "roco.pt").items() if k in model_dict}
    model_dict.update(pretrained)
    ref.load_state_dict(model_dict)
    torch.manual_seed(9)
    new_head = torch.nn.Linear(96, 17)
    ref.classifier[2] = torch.nn.Linear(96, 17)
    ref.classifier[2].load_state_dict(new_head.state_dict())
    # the build
    m = mmvqa_amd.Model(vqa_args)
    loaded, skipped, untouched = CK.load_roco_pretrained(m, str(tmp_path / "roco.pt"))
    assert skipped == ["head.0.bias", "head.0.weight", "head.2.bias", "head.2.weight"] and untouched == []
    m.classifier[2] = new_head
    sd, rsd = m.state_dict(), ref.state_dict()
    assert set(sd) == set(rsd)
    for k in rsd:
        assert torch.equal(sd[k], rsd[k]), k
    # strict load of the fine-tuned model (train.py:144 / eval.py:112), also from a recorder dict
    torch.save({"epoch": 3, "optimizer": {}, "scheduler": {}, "scaler": {}, "model": ref.state_dict()}, tmp_path / "recorder_2.pt")
    m2 = mmvqa_amd.Model(vqa_args)
    m2.classifier[2] = torch.nn.Linear(96, 17)
    CK.load_model(m2, str(tmp_path / "recorder_2.pt"))
    assert all(torch.equal(m2.state_dict()[k], rsd[k]) for k in rsd)
    with pytest.raises(RuntimeError):
        CK.load_model(mmvqa_amd.Model(vqa_args), str(tmp_path / "roco.pt"))      # strict: head.* is unexpected
