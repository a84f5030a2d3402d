#!/usr/bin/env python3
"""Generate the golden vectors that pin oracle/ against the reference.

Run ONCE in the build container (the reference never travels to the GPU box):

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports the reference's own modules from /root/reference on CPU (recipe:
SURVEY.md 8(c)), copies seeded weights from the oracle classes into them
(state_dict names are identical), runs the REFERENCE code and stores
inputs + reference outputs as small .npz fixtures next to this file.  Weights
are not stored: tests rebuild them from the recorded seed through the same
oracle constructors; a weight checksum in each fixture detects RNG drift.

Only data (inputs / expected outputs) is written -- no reference source text.
"""
import os
import sys
import types
import argparse

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import transformers  # noqa: F401  (must be imported before the stubs)
from oracle import mmbert_oracle as O  # noqa: E402
from oracle import effnet_oracle as E  # noqa: E402

# ---- stub the absent third-party packages (names only; no arithmetic in the stubs)
_tv = types.ModuleType("torchvision")
_tvm = types.ModuleType("torchvision.models")
_tvm.resnet152 = lambda **kw: None
_tv.models = _tvm
sys.modules["torchvision"] = _tv
sys.modules["torchvision.models"] = _tvm
_timm = types.ModuleType("timm")
_timm.create_model = lambda *a, **k: None
sys.modules["timm"] = _timm

import models.mmbert as RM  # noqa: E402
import models.image_encoding as RI  # noqa: E402
from models.transformer import BertLayer  # noqa: E402
from models.realformer import ResEncoderBlock  # noqa: E402
from models.serf import SERF  # noqa: E402
from models.asl_singlelabel import ASLSingleLabel  # noqa: E402
from models.SupConLoss.loss import SupConLoss  # noqa: E402
from transformers.models.bert.modeling_bert import BertEmbeddings  # noqa: E402
from transformers import BertConfig  # noqa: E402


def wsum(sd):
    return float(sum(v.double().abs().sum() for v in sd.values() if v.dtype.is_floating_point))


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: tuple(np.shape(v)) for k, v in out.items()})


def zero_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0


def ragged_mask(B, T, lens):
    m = torch.zeros(B, T, dtype=torch.long)
    for b, n in enumerate(lens):
        m[b, :n] = 1
    return m


# ----------------------------------------------------------------------------- per-op
def gold_activations():
    torch.manual_seed(11)
    x = torch.cat([torch.randn(500) * 4, torch.tensor([-100., -50., -20., 0., 20., 49.9, 50., 50.1, 80., 1e4])])
    x.requires_grad_(True)
    y = SERF()(x)
    y.sum().backward()
    from models.transformer import gelu
    x2 = x.detach().clone().requires_grad_(True)
    g = gelu(x2)
    g.sum().backward()
    save("act", x=x, serf=y, dserf=x.grad, gelu=g, dgelu=x2.grad)


def gold_bertlayer():
    seed, H, heads, L, B, T = 21, 96, 12, 2, 3, 10
    torch.manual_seed(seed)
    orc = O.OracleBertLayer(H, heads, L, 0.3)
    a = argparse.Namespace(hidden_size=H, heads=heads, n_layers=L, hidden_dropout_prob=0.3)
    ref = BertLayer(a, share="none", norm="pre")
    ref.load_state_dict(orc.state_dict())
    ref.eval()
    torch.manual_seed(seed + 1)
    x = torch.randn(B, T, H, requires_grad=True)
    mask = ragged_mask(B, T, [10, 7, 4])
    h = x
    for i in range(L):
        h = ref(h, mask, i)
    gy = torch.randn_like(h)
    (h * gy).sum().backward()
    grads = {("g_" + k.replace(".", "__")): (p.grad if p.grad is not None else torch.zeros_like(p))
             for k, p in ref.named_parameters()}
    save("bertlayer", seed=seed, dims=[H, heads, L, B, T], x=x, mask=mask, y=h, gy=gy, dx=x.grad,
         wsum=wsum(orc.state_dict()), **grads)


def gold_realformer():
    seed, emb_s, L, B, T = 31, 12, 3, 3, 9
    H = emb_s * 8
    torch.manual_seed(seed)
    orc = torch.nn.Sequential(*[O.OracleResEncoderBlock(emb_s, 8) for _ in range(L)])
    ref = torch.nn.Sequential(*[ResEncoderBlock(emb_s=emb_s, head_cnt=8) for _ in range(L)])
    ref.load_state_dict(orc.state_dict())
    ref.eval()
    torch.manual_seed(seed + 1)
    x = torch.randn(B, T, H, requires_grad=True)
    mask = ragged_mask(B, T, [9, 6, 3])
    h, prev = x, None
    for blk in ref:
        h, prev = blk(h, prev=prev, mask=mask)
    gy = torch.randn_like(h)
    (h * gy).sum().backward()
    grads = {("g_" + k.replace(".", "__")): p.grad for k, p in ref.named_parameters()}
    save("realformer", seed=seed, dims=[emb_s, L, B, T], x=x, mask=mask, y=h, prev=prev, gy=gy, dx=x.grad,
         wsum=wsum(orc.state_dict()), **grads)


def gold_losses():
    torch.manual_seed(41)
    logits = (torch.randn(6, 37) * 3).requires_grad_(True)
    tgt = torch.randint(0, 37, (6,))
    l = ASLSingleLabel()(logits, tgt)
    l.backward()
    feat = torch.nn.functional.normalize(torch.randn(5, 2, 16), dim=2).requires_grad_(True)
    ls = SupConLoss(temperature=0.07)(feat)
    ls.backward()
    # MLM loss exactly as the caller computes it (pretrain/roco_utils.py:235-236,257-265)
    lg = (torch.randn(2, 7, 50) * 2).requires_grad_(True)
    t = torch.tensor([[0, 0, 5, 0, 49, 0, 0], [0, 3, 0, 0, 0, 0, 0]])
    lp = lg.log_softmax(-1)
    lm = torch.nn.NLLLoss()(lp.permute(0, 2, 1), t)
    lm.backward()
    sel = t > 0
    pred = lp[sel, :].argmax(1)
    save("losses", asl_logits=logits, asl_target=tgt, asl=l, asl_dlogits=logits.grad,
         sc_feat=feat, sc=ls, sc_dfeat=feat.grad,
         mlm_logits=lg, mlm_target=t, mlm=lm, mlm_dlogits=lg.grad, mlm_pred=pred)


# ----------------------------------------------------------------------------- full model
def build_ref_model(args, orc):
    """Reference Model(args) with network fetches replaced:
    BertEmbeddings random-init with a small config, resnet152 -> OracleResNet mini."""
    cfg = BertConfig(vocab_size=args.vocab_size, hidden_size=args.hidden_size,
                     max_position_embeddings=args.bert_max_pos)
    RM.TransformerAbstract.get_bert_embedding = lambda self, a: BertEmbeddings(cfg)
    RI.models_dict[5]["resnet152"][0] = lambda pretrained=True: O.OracleResNet(args.resnet_layers, args.resnet_width)
    # timm.create_model(name, features_only=True, pretrained=True) (image_encoding.py:26): the reference's own
    # Timm_EFfNetV2.__init__/forward (image_encoding.py:89-115) then runs on the oracle's features-only body
    RI.models_dict[5]["tf_efficientnetv2_m"][0] = (
        lambda name, features_only=True, pretrained=True: E.OracleEffNetV2Features(getattr(args, "effnet_depth_div", 1)))
    ref = RM.Model(args)
    if "efficientnetv2" in args.cnn_encoder:
        assert type(ref.transformer.trans).__name__ == "Timm_EFfNetV2"
    missing = ref.load_state_dict(orc.state_dict(), strict=True)
    return ref


def gold_model(tag, transformer_model, dataset, supcon, B, T, img_hw, lens, cnn="resnet152", use_relu=False):
    seed = 51
    kw = dict(transformer_model=transformer_model, dataset=dataset, hidden_size=768, n_layers=2, heads=12,
              hidden_dropout_prob=0.0, vocab_size=64 if dataset == "roco" else 23,
              resnet_layers=(1, 1, 1, 1), resnet_width=64, bert_max_pos=32, use_relu=use_relu, cnn_encoder=cnn)
    if supcon:
        kw["supcon"] = True
    eff = "efficientnetv2" in cnn
    if eff:
        kw["effnet_depth_div"] = 8   # every stage type / width / stride / SE of the full body, 1-3 blocks per stage
    args = O.make_args(**kw)
    # vocab_size doubles as embedding vocab AND classifier width in the reference (mmbert.py:137)
    torch.manual_seed(seed)
    orc = O.OracleModel(args)
    ref = build_ref_model(args, orc)
    zero_dropout(ref)
    ref.train()  # train-mode BatchNorm (batch statistics), dropout p = 0
    torch.manual_seed(seed + 1)
    img = torch.rand(B, 3, img_hw, img_hw) * 2 - 1
    ids = torch.randint(1, args.vocab_size, (B, T))
    ids[:, 1:6] = 0
    seg = torch.zeros(B, T, dtype=torch.long)
    mask = ragged_mask(B, T, lens)
    for b, n in enumerate(lens):
        seg[b, 7:n] = 1
        ids[b, n:] = 0
    out = ref(img, ids, seg, mask)
    arrs = dict(seed=seed, img=img, ids=ids, seg=seg, mask=mask, wsum=wsum(orc.state_dict()),
                dims=[B, T, img_hw, args.vocab_size])
    if dataset == "roco":
        logits = out[0] if supcon else out
        tgt = torch.zeros(B, T, dtype=torch.long)
        tgt[0, 8] = 5
        tgt[1, 7] = 9
        lp = logits.log_softmax(-1)
        loss = torch.nn.NLLLoss()(lp.permute(0, 2, 1), tgt)
        if supcon:
            feat = out[1]
            fs = torch.cat([feat[:B // 2].unsqueeze(1), feat[B // 2:].unsqueeze(1)], dim=1)
            loss = loss + SupConLoss(temperature=0.07)(fs)
            arrs["feat"] = feat
        arrs.update(logits=logits, target=tgt, loss=loss)
    else:
        logits, z0, z1 = out
        assert z0 == 0 and z1 == 0
        tgt = torch.randint(0, args.vocab_size, (B,))
        loss = ASLSingleLabel()(logits, tgt)
        arrs.update(logits=logits, target=tgt, loss=loss)
    loss.backward()
    sd = dict(ref.named_parameters())
    pick = ["fc1.weight", "classifier.1.bias", "transformer.bert_embedding.word_embeddings.weight",
            "transformer.bert_embedding.LayerNorm.weight", "transformer.trans.conv7.weight",
            "transformer.trans.conv2.weight", "transformer.trans.model.bn1.weight"]
    if eff:
        pick += ["transformer.trans.conv3.weight", "transformer.trans.conv4.weight", "transformer.trans.conv5.weight",
                 "transformer.trans.model.conv_stem.weight", "transformer.trans.model.blocks.0.0.conv.weight",
                 "transformer.trans.model.blocks.1.0.conv_exp.weight", "transformer.trans.model.blocks.2.0.conv_pwl.weight",
                 "transformer.trans.model.blocks.3.0.conv_dw.weight", "transformer.trans.model.blocks.3.0.se.conv_reduce.bias",
                 "transformer.trans.model.blocks.4.1.se.conv_expand.weight", "transformer.trans.model.blocks.5.0.bn2.bias",
                 "transformer.trans.model.blocks.6.0.conv_pw.weight"]
    else:
        pick += ["transformer.trans.model.conv1.weight", "transformer.trans.model.layer3.0.conv2.weight",
                 "transformer.trans.model.layer1.0.downsample.0.weight"]
    if transformer_model == "transformer":
        pick += ["transformer.blocks.norm1.weight", "transformer.blocks.attention.0.proj_q.weight"]
    else:
        pick += ["transformer.mains.0.kqv.weight", "transformer.mains.1.ln2.bias"]
    for k in pick:
        g = sd[k].grad
        arrs["g_" + k.replace(".", "__")] = g if g.numel() <= 8192 else g.flatten()[:: max(1, g.numel() // 4096)][:4096]
    # gradient "fingerprint" of every parameter: (sum, abs-sum)
    names, fp = [], []
    for k, p in sd.items():
        names.append(k)
        fp.append([0.0, 0.0] if p.grad is None else [float(p.grad.double().sum()), float(p.grad.double().abs().sum())])
    arrs["grad_names"] = np.array(names)
    arrs["grad_fp"] = np.array(fp)
    # BN running stats after ONE reference forward (quirk 7: k-fold updates)
    bsd = ref.state_dict()
    bkeys = ["transformer.trans.model.bn1.running_mean", "transformer.trans.model.bn1.running_var",
             "transformer.trans.model.bn1.num_batches_tracked"]
    if eff:   # one backbone pass (image_encoding.py:101) => every BatchNorm updates exactly once
        bkeys += ["transformer.trans.model.blocks.1.0.bn1.running_var", "transformer.trans.model.blocks.3.0.bn2.running_mean",
                  "transformer.trans.model.blocks.4.0.bn2.running_var", "transformer.trans.model.blocks.4.0.bn2.num_batches_tracked",
                  "transformer.trans.model.blocks.6.0.bn3.running_mean", "transformer.trans.model.blocks.6.0.bn3.num_batches_tracked"]
    else:
        bkeys += ["transformer.trans.model.layer2.0.bn2.running_var",
                  "transformer.trans.model.layer2.0.bn2.num_batches_tracked",
                  "transformer.trans.model.layer4.0.bn3.running_mean",
                  "transformer.trans.model.layer4.0.bn3.num_batches_tracked"]
    for k in bkeys:
        arrs["b_" + k.replace(".", "__")] = bsd[k]
    save(tag, **arrs)


EFF = "tf_efficientnetv2_m"
JOBS = {
    "act": gold_activations, "bertlayer": gold_bertlayer, "realformer": gold_realformer, "losses": gold_losses,
    "model_tr_roco": lambda: gold_model("model_tr_roco", "transformer", "roco", False, 2, 12, 64, [12, 9]),
    "model_rf_roco_supcon": lambda: gold_model("model_rf_roco_supcon", "realformer", "roco", True, 4, 11, 64, [11, 8, 10, 9]),
    "model_tr_vqa": lambda: gold_model("model_tr_vqa", "transformer", "VQA-Med", False, 3, 10, 64, [10, 8, 9]),
    "model_rf_vqa": lambda: gold_model("model_rf_vqa", "realformer", "VQA-Med", False, 2, 10, 64, [10, 9]),
    # the reference's Timm_EFfNetV2 (image_encoding.py:89-115) on the oracle's features-only body:
    # BASELINE configs[2] (MLM), configs[3] (MLM + SupCon head, 2N views), configs[4] (VQA head + ASL) shapes,
    # and the --use_relu tap variant (README rows "EfficientNetV2 + Transformer, ReLU")
    "model_eff_rf_roco": lambda: gold_model("model_eff_rf_roco", "realformer", "roco", False, 3, 12, 64, [12, 9, 10], cnn=EFF),
    "model_eff_rf_roco_supcon": lambda: gold_model("model_eff_rf_roco_supcon", "realformer", "roco", True, 4, 11, 64,
                                                   [11, 8, 10, 9], cnn=EFF),
    "model_eff_rf_vqa_asl": lambda: gold_model("model_eff_rf_vqa_asl", "realformer", "VQA-Med", False, 4, 10, 72,
                                               [10, 8, 9, 10], cnn=EFF),
    "model_eff_tr_roco_relu": lambda: gold_model("model_eff_tr_roco_relu", "transformer", "roco", False, 2, 12, 64, [12, 9],
                                                 cnn=EFF, use_relu=True),
}

if __name__ == "__main__":
    for name in (sys.argv[1:] or list(JOBS)):   # no arguments: regenerate everything
        JOBS[name]()
