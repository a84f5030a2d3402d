"""CPU restatement (pure torch ops, fp32) of the MM-VQA MMBERT hot path.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  The product package never
imports this file; the HIP path fails loudly when its extension is missing.

Every class/function cites the reference file:line it restates (paths are
relative to the upstream repository DannielSilva/MM-VQA).  Parity status:

* transformer / realformer / SERF / heads / mean_pooling / ASL / SupCon /
  prepare_input / Model.forward: PINNED -- tests/golden/make_golden.py runs the
  reference's own modules on CPU with weights copied from these classes and
  stores inputs+outputs; tests/test_oracle_golden.py replays them.
* ResNet-152 / EfficientNetV2-M graphs: the reference takes them from
  torchvision / timm, neither of which exists in the build image, and the
  reference has no tests => "parity unpinned" for the backbone *wiring*.
  The arithmetic (conv2d / batch_norm on CPU) is torch's; the wiring is checked
  by parameter counts (60 192 808 for ResNet-152), tap shapes and
  torchvision-compatible state_dict names.  The reference's own tap code
  (image_encoding.py:71-87) IS pinned: the golden script runs it on top of
  OracleResNet (children() order preserved) and compares with the single-pass
  restatement below.

State-dict key names equal the reference's, so weights move
between reference <-> oracle <-> HIP model with load_state_dict.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- activations
def gelu(x):
    """models/transformer.py:7-8 -- exact erf GELU."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def serf(x, thresh: float = 50.0):
    """models/serf.py:23-24 -- x * erf(log1p(exp(min(x, 50))))."""
    return x * torch.erf(torch.log1p(torch.exp(torch.clamp(x, max=thresh))))


class Serf(nn.Module):
    def forward(self, x):
        return serf(x)


# --------------------------------------------------------------------------- embeddings
class OracleBertEmbeddings(nn.Module):
    """HF transformers BertEmbeddings as called at models/mmbert.py:63 with
    position_ids=None: LN((word[ids] + type[seg]) + pos[0:T]; eps 1e-12) then
    dropout(0.1).  (transformers/models/bert/modeling_bert.py, 5.x: 53-108.)"""

    def __init__(self, vocab_size=30522, hidden_size=768, max_position_embeddings=512,
                 type_vocab_size=2, layer_norm_eps=1e-12, hidden_dropout_prob=0.1, pad_token_id=0):
        super().__init__()
        self.word_embeddings = nn.Embedding(vocab_size, hidden_size, padding_idx=pad_token_id)
        self.position_embeddings = nn.Embedding(max_position_embeddings, hidden_size)
        self.token_type_embeddings = nn.Embedding(type_vocab_size, hidden_size)
        self.LayerNorm = nn.LayerNorm(hidden_size, eps=layer_norm_eps)
        self.dropout = nn.Dropout(hidden_dropout_prob)
        # HF registers these as non-persistent buffers -> not in state_dict
        self.register_buffer("position_ids", torch.arange(max_position_embeddings).expand((1, -1)),
                             persistent=False)

    def forward(self, input_ids, token_type_ids, position_ids=None):
        T = input_ids.shape[1]
        e = self.word_embeddings(input_ids) + self.token_type_embeddings(token_type_ids)
        e = e + self.position_embeddings(self.position_ids[:, :T])
        return self.dropout(self.LayerNorm(e))


# --------------------------------------------------------------------------- ResNet (torchvision layout)
class OracleBottleneck(nn.Module):
    """torchvision Bottleneck v1.5 (stride on the 3x3) -- SURVEY.md Appendix A."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        idt = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            idt = self.downsample(x)
        return self.relu(out + idt)


class OracleResNet(nn.Module):
    """torchvision ResNet with children() order
    [conv1, bn1, relu, maxpool, layer1..4, avgpool, fc] so that the reference's
    slicing (models/image_encoding.py:72-85) works on it.  layers=(3,8,36,3),
    width=64 is resnet152 (60 192 808 parameters)."""

    def __init__(self, layers=(3, 8, 36, 3), width=64, num_classes=1000):
        super().__init__()
        self.inplanes = width
        self.conv1 = nn.Conv2d(3, width, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(width, layers[0], 1)
        self.layer2 = self._make_layer(width * 2, layers[1], 2)
        self.layer3 = self._make_layer(width * 4, layers[2], 2)
        self.layer4 = self._make_layer(width * 8, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(width * 32, num_classes)
        for m in self.modules():  # torchvision init
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, planes, blocks, stride):
        ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                           nn.BatchNorm2d(planes * 4))
        layers = [OracleBottleneck(self.inplanes, planes, stride, ds)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(OracleBottleneck(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def resnet152(**_):
    return OracleResNet((3, 8, 36, 3), 64)


def _run_bn_reps(mod: nn.Module, x: torch.Tensor, reps: int):
    """Reference quirk 7: the ResNet prefix is executed once per
    tap, so a BatchNorm that sits under k taps updates its running stats k times
    with the *same* batch statistics: r <- (1-m)^k r + (1-(1-m)^k) s and
    num_batches_tracked += k.  Done here as ONE update with the effective
    momentum 1-(1-m)^k (running stats must not be touched again after the
    forward: autograd holds them)."""
    bns = [m for m in mod.modules() if isinstance(m, nn.BatchNorm2d)]
    train = any(b.training for b in bns)
    if not train or reps == 1:
        return mod(x)
    old = [b.momentum for b in bns]
    for b in bns:
        b.momentum = 1.0 - (1.0 - b.momentum) ** reps
    try:
        y = mod(x)
    finally:
        for b, m in zip(bns, old):
            b.momentum = m
    with torch.no_grad():
        for b in bns:
            b.num_batches_tracked += reps - 1
    return y


class OracleResNetTransfer(nn.Module):
    """models/image_encoding.py:43-87 (Transfer + ResNetTransfer), restated as ONE
    backbone pass with five taps.  Forward values and gradients equal the
    reference's five prefix passes; BN running stats follow the k-fold rule
    (stem 5x, layer1 4x, layer2 3x, layer3 2x, layer4 1x).
    Returns (v_2, v_3, v_4, v_5, v_7) = taps of (layer4, layer3, layer2, layer1, stem)."""

    def __init__(self, hidden_size=768, use_relu=False, layers=(3, 8, 36, 3), width=64):
        super().__init__()
        self.model = OracleResNet(layers, width)
        cs = [width * 32, width * 16, width * 8, width * 4, width]
        self.channel_size = cs
        self.hidden_size = hidden_size
        self.use_relu = use_relu
        for name, c in zip(("conv2", "conv3", "conv4", "conv5", "conv7"), cs):
            setattr(self, name, nn.Conv2d(c, hidden_size, kernel_size=1, stride=1, bias=False))

    def _act(self, x):
        return F.relu(x) if self.use_relu else serf(x)

    def _tap(self, conv, fmap):
        return self._act(conv(fmap)).mean(dim=(2, 3)).view(-1, self.hidden_size)

    def _run(self, mod, x, reps):
        return _run_bn_reps(mod, x, reps)

    def forward(self, img):
        m = self.model
        stem = self._run(nn.Sequential(m.conv1, m.bn1, m.relu), img, 5)
        v_7 = self._tap(self.conv7, stem)
        l1 = self._run(m.layer1, m.maxpool(stem), 4)
        v_5 = self._tap(self.conv5, l1)
        l2 = self._run(m.layer2, l1, 3)
        v_4 = self._tap(self.conv4, l2)
        l3 = self._run(m.layer3, l2, 2)
        v_3 = self._tap(self.conv3, l3)
        l4 = self._run(m.layer4, l3, 1)
        v_2 = self._tap(self.conv2, l4)
        return v_2, v_3, v_4, v_5, v_7


# --------------------------------------------------------------------------- BertLayer stack
class _MHSA(nn.Module):
    """models/transformer.py:10-40."""

    def __init__(self, hidden, heads, p):
        super().__init__()
        self.proj_q = nn.Linear(hidden, hidden)
        self.proj_k = nn.Linear(hidden, hidden)
        self.proj_v = nn.Linear(hidden, hidden)
        self.drop = nn.Dropout(p)
        self.n_heads = heads

    def forward(self, x, mask):
        B, T, H = x.shape
        h = self.n_heads
        q, k, v = (p(x).view(B, T, h, H // h).transpose(1, 2) for p in (self.proj_q, self.proj_k, self.proj_v))
        scores = q @ k.transpose(-2, -1) / float(math.sqrt(H // h))
        if mask is not None:
            scores = scores - 10000.0 * (1.0 - mask[:, None, None, :].float())
        scores = self.drop(F.softmax(scores, dim=-1))
        return (scores @ v).transpose(1, 2).contiguous().view(B, T, H)


class _FFN(nn.Module):
    """models/transformer.py:42-48."""

    def __init__(self, hidden):
        super().__init__()
        self.fc1 = nn.Linear(hidden, hidden * 4)
        self.fc2 = nn.Linear(hidden * 4, hidden)

    def forward(self, x):
        return self.fc2(gelu(self.fc1(x)))


class OracleBertLayer(nn.Module):
    """models/transformer.py:50-98 with share='none', norm='pre' (the only mode
    models/mmbert.py:87 uses).  Quirk: norm1 is used for BOTH sub-layers of EVERY
    layer; norm2 exists as parameters but is never applied."""

    def __init__(self, hidden, heads, n_layers, p):
        super().__init__()
        self.norm1 = nn.LayerNorm(hidden, eps=1e-12)
        self.norm2 = nn.LayerNorm(hidden, eps=1e-12)
        self.drop1 = nn.Dropout(p)
        self.drop2 = nn.Dropout(p)
        self.attention = nn.ModuleList([_MHSA(hidden, heads, p) for _ in range(n_layers)])
        self.proj = nn.ModuleList([nn.Linear(hidden, hidden) for _ in range(n_layers)])
        self.feedforward = nn.ModuleList([_FFN(hidden) for _ in range(n_layers)])

    def forward(self, x, mask, i):
        h = self.proj[i](self.attention[i](self.norm1(x), mask))
        out = x + self.drop1(h)
        h = self.feedforward[i](self.norm1(out))
        return out + self.drop2(h)


# --------------------------------------------------------------------------- RealFormer
class OracleResEncoderBlock(nn.Module):
    """models/realformer.py:9-51.  One kqv weight [3*emb_s, emb_s] shared by all
    heads, split order k,q,v; residual scores in layout [B, Ti, Tj, h]; the
    mask is applied along the QUERY axis (quirk 3); post-LN eps 1e-5; SERF FFN."""

    def __init__(self, emb_s=96, head_cnt=8, dp1=0.1, dp2=0.1):
        super().__init__()
        emb = emb_s * head_cnt
        self.kqv = nn.Linear(emb_s, 3 * emb_s, bias=False)
        self.dp = nn.Dropout(dp1)
        self.proj = nn.Linear(emb, emb, bias=False)
        self.head_cnt, self.emb_s = head_cnt, emb_s
        self.ln1 = nn.LayerNorm(emb)
        self.ln2 = nn.LayerNorm(emb)
        self.ff = nn.Sequential(nn.Linear(emb, 4 * emb), Serf(), nn.Linear(4 * emb, emb), nn.Dropout(dp2))

    def resmha(self, x, prev, mask):
        B, T, _ = x.shape
        x = x.reshape(B, T, self.head_cnt, self.emb_s)
        k, q, v = torch.split(self.kqv(x), self.emb_s, dim=-1)
        att = torch.einsum("bihk,bjhk->bijh", q, k) / self.emb_s ** 0.5
        if prev is not None:
            att = att + prev
        if mask is not None:
            m = mask.unsqueeze(-1).unsqueeze(-1).expand(att.size()).float()
            att = att - 10000.0 * (1.0 - m)
        prev = att
        p = F.softmax(prev, dim=2)
        res = torch.einsum("btih,bihs->bths", p, v).reshape(B, T, -1)
        return self.dp(self.proj(res)), prev

    def forward(self, x, prev=None, mask=None):
        r, prev = self.resmha(x, prev, mask)
        x = self.ln1(x + r)
        x = self.ln2(x + self.ff(x))
        return x, prev


# --------------------------------------------------------------------------- fusion encoders
class _Abstract(nn.Module):
    """models/mmbert.py:45-67 (TransformerAbstract)."""

    def __init__(self, args):
        super().__init__()
        self.bert_embedding = OracleBertEmbeddings(
            vocab_size=getattr(args, "emb_vocab", getattr(args, "vocab_size", 30522)), hidden_size=args.hidden_size,
            max_position_embeddings=getattr(args, "bert_max_pos", 512))
        if "resnet" in args.cnn_encoder:
            self.trans = OracleResNetTransfer(args.hidden_size, getattr(args, "use_relu", False),
                                              getattr(args, "resnet_layers", (3, 8, 36, 3)),
                                              getattr(args, "resnet_width", 64))
        elif "efficientnetv2" in args.cnn_encoder:
            from oracle.effnet_oracle import OracleTimmEffNetV2
            self.trans = OracleTimmEffNetV2(args.hidden_size, getattr(args, "use_relu", False),
                                            getattr(args, "effnet_depth_div", 1))
        else:
            raise NotImplementedError

    def prepare_input(self, img, input_ids, token_type_ids, mask):
        """mmbert.py:60-67: rows 0..num_vis-1 of every sample are overwritten by the
        visual tokens AFTER embedding LayerNorm+dropout (quirk 1)."""
        vizs = list(self.trans(img))
        h = self.bert_embedding(input_ids, token_type_ids)
        h = h.clone()
        for n, v in enumerate(vizs):
            h[:, n, :] = v
        return h


class OracleTransformer(_Abstract):
    """models/mmbert.py:83-94."""

    def __init__(self, args):
        super().__init__(args)
        self.blocks = OracleBertLayer(args.hidden_size, args.heads, args.n_layers, args.hidden_dropout_prob)
        self.n_layers = args.n_layers

    def forward(self, img, input_ids, token_type_ids, mask):
        h = self.prepare_input(img, input_ids, token_type_ids, mask)
        for i in range(self.n_layers):
            h = self.blocks(h, mask, i)
        return h


class OracleRealFormer(_Abstract):
    """models/mmbert.py:96-108 (head_cnt hard-coded to 8)."""

    def __init__(self, args):
        super().__init__(args)
        self.mains = nn.Sequential(*[OracleResEncoderBlock(args.hidden_size // 8, 8, 0.1, 0.1)
                                     for _ in range(args.n_layers)])

    def forward(self, img, input_ids, token_type_ids, mask):
        h = self.prepare_input(img, input_ids, token_type_ids, mask)
        prev = None
        for blk in self.mains:
            h, prev = blk(h, prev=prev, mask=mask)
        return h


def mean_pooling(token_embeddings, attention_mask):
    """models/mmbert.py:169-172."""
    m = attention_mask.unsqueeze(-1).expand(token_embeddings.size()).float()
    return torch.sum(token_embeddings * m, 1) / torch.clamp(m.sum(1), min=1e-9)


class OracleModel(nn.Module):
    """models/mmbert.py:129-167."""

    def __init__(self, args, feat_dim=128):
        super().__init__()
        if "realformer" in args.transformer_model:
            self.transformer = OracleRealFormer(args)
        elif "transformer" in args.transformer_model:
            self.transformer = OracleTransformer(args)
        else:
            raise NotImplementedError
        H = args.hidden_size
        self.fc1 = nn.Linear(H, H)
        self.classifier = nn.Sequential(nn.Linear(H, H), nn.LayerNorm(H, eps=1e-12), nn.Linear(H, args.vocab_size))
        self.task, self.dataset = args.task, args.dataset
        self.supcon = getattr(args, "supcon", False)
        if self.supcon:
            self.head = nn.Sequential(nn.Linear(H, H), Serf(), nn.Linear(H, feat_dim))

    def forward(self, img, input_ids, segment_ids, input_mask):
        h = self.transformer(img, input_ids, segment_ids, input_mask)
        if self.dataset == "roco":
            logits = self.classifier(serf(self.fc1(h)))
            if self.supcon:
                feat = F.normalize(self.head(mean_pooling(h, input_mask)), dim=1)
                return logits, feat
            return logits
        elif self.dataset == "VQA-Med":
            logits = self.classifier(serf(self.fc1(mean_pooling(h, input_mask))))
            return logits, 0, 0
        raise NotImplementedError


# --------------------------------------------------------------------------- losses
def mlm_loss(logits, target):
    """pretrain/roco_utils.py:235-236: log_softmax + NLLLoss() over ALL B*T positions
    (label 0 is NOT ignored, quirk 5)."""
    lp = logits.log_softmax(-1)
    return F.nll_loss(lp.permute(0, 2, 1), target), lp


def mlm_accuracy(lp, target):
    """pretrain/roco_utils.py:257-265: argmax over positions with target > 0.
    Returns (pred[int64 n_masked], n_correct, n_masked)."""
    sel = target > 0
    pred = lp[sel, :].argmax(1)
    return pred, int((pred == target[sel]).sum()), int(sel.sum())


def asl_single_label(logits, target, gamma_pos=0.0, gamma_neg=4.0, eps=0.1):
    """models/asl_singlelabel.py:23-53."""
    C = logits.size(-1)
    lp = F.log_softmax(logits, dim=-1)
    t = torch.zeros_like(logits).scatter_(1, target.long().unsqueeze(1), 1)
    anti = 1 - t
    xs_pos = torch.exp(lp)
    xs_neg = 1 - xs_pos
    w = torch.pow(1 - xs_pos * t - xs_neg * anti, gamma_pos * t + gamma_neg * anti)
    lpw = lp * w
    ts = t * (1 - eps) + eps / C
    return (-(ts * lpw).sum(-1)).mean()


def supcon_simclr(features, temperature=0.07, base_temperature=0.07):
    """models/SupConLoss/loss.py:21-98 as called at supcon_utils.py:287 (features only
    => SimCLR).  features [N, 2, D]."""
    N = features.shape[0]
    f = torch.cat(torch.unbind(features, dim=1), dim=0)
    z = (f @ f.T) / temperature
    z = z - z.max(dim=1, keepdim=True)[0].detach()
    lm = 1.0 - torch.eye(2 * N, dtype=f.dtype)
    pos = torch.eye(N, dtype=f.dtype).repeat(2, 2) * lm
    logp = z - torch.log((torch.exp(z) * lm).sum(1, keepdim=True))
    mlpp = (pos * logp).sum(1) / pos.sum(1)
    return (-(temperature / base_temperature) * mlpp).view(2, N).mean()


def split_feat(feat, bsz):
    """models/SupConLoss/supcon_utils.py:259-261."""
    f1, f2 = torch.split(feat, [bsz, bsz], dim=0)
    return torch.cat([f1.unsqueeze(1), f2.unsqueeze(1)], dim=1)


# --------------------------------------------------------------------------- optimizer
def adam_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (pretrain/roco_train.py:90), single tensor, in place."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


# --------------------------------------------------------------------------- helpers
def make_args(**kw):
    """Namespace with the fields Model(args) reads (SURVEY.md section 5 'Config')."""
    d = dict(task="MLM", dataset="roco", transformer_model="transformer", cnn_encoder="resnet152",
             num_vis=5, hidden_size=768, n_layers=4, heads=12, hidden_dropout_prob=0.3,
             vocab_size=30522, use_relu=False, max_position_embeddings=32)
    d.update(kw)
    return SimpleNamespace(**d)
