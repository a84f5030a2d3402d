"""GPU parity of the whole hot path: mmvqa_amd.Model (HIP engine behind the reference's
Model(args) protocol) against the CPU oracle on identical weights/inputs, plus direct replay of
the golden vectors the REFERENCE produced.  Tolerance: 1e-3 relative to each tensor's max
(BASELINE.json north_star: "within 1e-3 fp32"); index outputs bit-exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import mmvqa_amd  # noqa: E402
from mmvqa_amd import synth  # noqa: E402
from oracle import mmbert_oracle as O  # noqa: E402
from hip_helpers import dev, relerr  # noqa: E402
from test_oracle_golden import MODEL_CASES, model_case_args  # noqa: E402

TOL = 1e-3


def zero_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0


def mini_args(**kw):
    d = dict(resnet_layers=(1, 1, 1, 1), resnet_width=8, hidden_size=96, n_layers=2, heads=12, vocab_size=50,
             emb_vocab=50, bert_max_pos=32, hidden_dropout_prob=0.0, emb_dropout_prob=0.0, rf_dropout_prob=0.0)
    d.update(kw)
    return O.make_args(**d)


def build_pair(args, seed=0):
    torch.manual_seed(seed)
    oargs = O.make_args(**{**vars(args), "vocab_size": args.vocab_size})
    orc = O.OracleModel(oargs)
    # randomise BN affine / running stats so that mistakes there are visible
    with torch.no_grad():
        for m in orc.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
    zero_dropout(orc)
    hip = mmvqa_amd.Model(args)
    hip.load_state_dict(orc.state_dict())
    hip.to(dev())
    return orc, hip


def compare_grads(orc, hip, orc64=None, tol=TOL):
    """HIP gradients vs the oracle.  Where an fp64 run of the oracle is given, it is the truth and the
    tolerance of a tensor is max(tol, 5 x the fp32 oracle's own distance from it): some gradients are
    ill-conditioned in fp32 on tiny batches (k-bias is identically 0 in exact arithmetic; BatchNorm over
    16 samples; RealFormer's -10000*k score offsets) and the reference's fp32 result carries that noise."""
    bad = []
    hp = dict(hip.named_parameters())
    p64 = dict(orc64.named_parameters()) if orc64 is not None else None
    for name, p in orc.named_parameters():
        g_ref = p.grad
        g = hp[name].grad
        if g_ref is None:
            assert g is None or float(g.abs().max()) == 0.0, f"{name}: reference has no gradient"
            continue
        assert g is not None, f"{name}: missing gradient"
        t = tol
        if p64 is not None:
            truth = p64[name].grad
            t = max(tol, 5 * relerr(g_ref, truth))
            g_ref = truth
        e = relerr(g, g_ref)
        if not e <= t:
            bad.append((name, e, t))
    assert not bad, "gradient mismatches: " + ", ".join(f"{n} {e:.2e} (tol {t:.1e})" for n, e, t in bad[:12])


def oracle_loss(kind, out, tgt, B):
    if kind == "vqa":
        return O.asl_single_label(out[0], tgt)
    if kind == "supcon":
        return O.mlm_loss(out[0], tgt)[0] + O.supcon_simclr(O.split_feat(out[1], B // 2))
    return O.mlm_loss(out, tgt)[0]


def run_case(args, B, T, hw, kind, seed=0, stat_tol=1e-4, tune=False):
    """tune=True: the launcher's timed per-shape choices (tile, split-K, K split + finishing launch, persistent form) are
    made first -- Model.tune() on the case's own inputs, as bench.py and train.py do -- so that the kernels compared
    with the oracle are the ones the timed steps run.  tune="both": the default launch choices AND the tuned ones against
    ONE evaluation of the CPU oracle (fp32 and fp64: the expensive part of a full-size case).  tune="all": a third model
    whose tuning pass runs with MMVQA_PERSIST_KINDS=7 (bits 0/1 let the tuner try the persistent stream-K
    form for forward / data-gradient and weight-gradient products, bit 2 makes it win wherever it runs), so that every
    eligible product of the step -- both streams' ticketed fix-ups included -- runs in that form."""
    orc, hip = build_pair(args, seed)
    V = args.vocab_size
    if kind == "vqa":
        img, ids, seg, mask, tgt = synth.vqa_batch(B, T, hw, vocab=args.emb_vocab, n_classes=V, seed=5)
    else:
        img, ids, seg, mask, tgt = synth.roco_batch(B, T, hw, vocab=V, seed=5, mlm_prob=0.3)
    import copy
    init_sd = {k: v.detach().clone() for k, v in orc.state_dict().items()}
    orc64 = copy.deepcopy(orc).double().train()
    orc.train()
    out_ref = orc(img, ids, seg, mask)
    loss_ref = oracle_loss(kind, out_ref, tgt, B)
    loss_ref.backward()
    oracle_loss(kind, orc64(img.double(), ids, seg, mask), tgt, B).backward()
    osd = orc.state_dict()
    dimg, dids, dseg, dmask, dtgt = (t.to(dev()) for t in (img, ids, seg, mask, tgt))
    for tuned in {"both": (False, True), "all": (False, True, "persist")}.get(tune, (bool(tune),)):
        if hip is None:
            hip = mmvqa_amd.Model(args)
            hip.load_state_dict(init_sd)
            hip.to(dev())
        hip.train()
        if tuned == "persist":
            os.environ["MMVQA_PERSIST_KINDS"] = "7"
            try:
                n = hip.tune(dimg, dids, dseg, dmask)
            finally:
                del os.environ["MMVQA_PERSIST_KINDS"]
            assert hip.tuned_persistent() >= 20, (hip.tuned_persistent(), n)
        elif tuned:
            n = hip.tune(dimg, dids, dseg, dmask)
            assert n > 20 and hip.tuned_persistent() == 0, (n, hip.tuned_persistent())
        what = {False: "", True: "tuned launches: ", "persist": "persistent form: "}[tuned]
        out = hip(dimg, dids, dseg, dmask)
        if kind == "vqa":
            assert out[1] == 0 and out[2] == 0
            logits, logits_ref = out[0], out_ref[0]
            loss = mmvqa_amd.asl_loss(logits, dtgt)
        elif kind == "supcon":
            logits, feat = out
            logits_ref, feat_ref = out_ref
            assert relerr(feat, feat_ref) <= TOL, f"{what}feat {relerr(feat, feat_ref):.2e}"
            loss = mmvqa_amd.mlm_loss(logits, dtgt)[0] + mmvqa_amd.supcon_loss(mmvqa_amd.split_feat(feat, B // 2))
        else:
            logits, logits_ref = out, out_ref
            loss = mmvqa_amd.mlm_loss(logits, dtgt)[0]
        e = relerr(logits, logits_ref)
        assert e <= TOL, f"{what}logits rel err {e:.2e}"
        assert abs(float(loss) - float(loss_ref)) <= TOL * abs(float(loss_ref)), (what, float(loss), float(loss_ref))
        loss.backward()
        compare_grads(orc, hip, orc64)
        # BatchNorm running statistics incl. the k-fold update rule (quirk 7)
        hsd = hip.state_dict()
        for k, v in osd.items():
            if "running_" in k:
                assert relerr(hsd[k], v) <= stat_tol, f"{what}{k}: {relerr(hsd[k], v):.2e}"
            if k.endswith("num_batches_tracked"):
                assert int(hsd[k]) == int(v), k
        last, hip = hip, None
    return orc, last


@pytest.mark.parametrize("tm", ["transformer", "realformer"])
def test_mlm_mini(tm):
    run_case(mini_args(transformer_model=tm), B=3, T=12, hw=32, kind="mlm")


def test_mlm_supcon_mini():
    run_case(mini_args(transformer_model="realformer", supcon=True), B=4, T=11, hw=32, kind="supcon")


@pytest.mark.parametrize("tm", ["transformer", "realformer"])
def test_vqa_mini(tm):
    run_case(mini_args(transformer_model=tm, dataset="VQA-Med", vocab_size=23), B=3, T=10, hw=32, kind="vqa")


def test_deeper_backbone_relu_taps():
    """identity blocks, several blocks per layer, --use_relu taps, T=32, odd image size"""
    run_case(mini_args(resnet_layers=(2, 2, 3, 2), resnet_width=16, use_relu=True), B=2, T=32, hw=72, kind="mlm")


@pytest.mark.parametrize("tm,hw", [("realformer", 64), ("transformer", 72)])
def test_effnetv2_backbone(tm, hw):
    """tf_efficientnetv2_m body (depth reduced 8x; widths, SE, depthwise, SAME padding as the full net)"""
    run_case(mini_args(cnn_encoder="tf_efficientnetv2_m", effnet_depth_div=8, transformer_model=tm), B=3, T=12, hw=hw,
             kind="mlm")


def test_effnetv2_vqa_relu_taps():
    run_case(mini_args(cnn_encoder="tf_efficientnetv2_m", effnet_depth_div=5, dataset="VQA-Med", vocab_size=23,
                       use_relu=True), B=4, T=10, hw=64, kind="vqa")


def test_full_width_hidden768():
    """real channel widths (64..2048) and hidden 768 with a one-block-per-layer backbone"""
    run_case(mini_args(resnet_width=64, hidden_size=768, n_layers=1, vocab_size=300, emb_vocab=300), B=2, T=32,
             hw=64, kind="mlm")


def test_full_config2_resnet152_224():
    """BASELINE.json configs[1] itself: resnet152 at full depth and width, 224x224, hidden 768, 4 layers, T 32,
    vocab 30522 -- forward, loss, every gradient and the BatchNorm buffers against the oracle (batch 2 keeps the
    CPU oracle, fp32 and fp64, within seconds)"""
    # (running variances 100 BatchNorms deep carry the fp32 noise of everything below them: 1e-3 like the outputs)
    run_case(O.make_args(hidden_dropout_prob=0.0, emb_dropout_prob=0.0, rf_dropout_prob=0.0), B=2, T=32, hw=224,
             kind="mlm", stat_tol=TOL)


def test_full_config2_batch16_the_bench_shape():
    """configs[1] exactly as bench.py runs it: per-GPU batch 16 (the tile / split-K choices and the grids differ from
    the batch-2 case above)"""
    # ... first with the launcher's default choices, then through the TUNED launcher (Model.tune(), as bench.py does):
    # what bench.py times is what the oracle checks; then with every eligible product in the opt-in persistent form.
    # (The persistent form is checked at THIS size on purpose: on a few-pixel mini network one ReLU within fp32 rounding
    # of its kink moves a gradient by 1e-2 -- for the fp32 oracle against its own fp64 run just as for any kernel whose
    # summation order differs -- and a different order flips different ones.)
    run_case(O.make_args(hidden_dropout_prob=0.0, emb_dropout_prob=0.0, rf_dropout_prob=0.0), B=16, T=32, hw=224,
             kind="mlm", stat_tol=TOL, tune="all")


def test_full_config1_resnet152_transformer_vqa_head():
    """BASELINE.json configs[0] (vqamed2019/train.py:117-160 construction): resnet152 at full depth + transformer + the VQA
    head (masked mean pooling, classifier over the answer classes), batch 4, T 28 (the script's default), 224x224,
    CrossEntropy-free comparison of logits plus the ASL loss / gradients; default and tuned launch choices"""
    a = O.make_args(dataset="VQA-Med", vocab_size=1552, emb_vocab=30522, hidden_dropout_prob=0.0, emb_dropout_prob=0.0,
                    rf_dropout_prob=0.0)
    run_case(a, B=4, T=28, hw=224, kind="vqa", stat_tol=TOL, tune="both")


def test_full_config3_and_4_effnetv2m_realformer_mlm_supcon_224():
    """BASELINE.json configs[2] AND configs[3]: tf_efficientnetv2_m at full depth (57 blocks) + RealFormer, 224x224, T 32,
    vocab 30522 -- configs[2] is pretrain/roco_train.py (MLM head), configs[3] is pretrain/roco_supcon_train.py, the same
    model with the SupCon head added on 2N views (two crops per sample concatenated along the batch): one run checks
    the MLM logits and loss of both, feat, the SupCon loss, every gradient and the BatchNorm buffers against the fp32 /
    fp64 oracle (2N = 8 views)"""
    run_case(O.make_args(cnn_encoder="tf_efficientnetv2_m", transformer_model="realformer", heads=8, supcon=True,
                         hidden_dropout_prob=0.0, emb_dropout_prob=0.0, rf_dropout_prob=0.0), B=8, T=32, hw=224,
             kind="supcon", stat_tol=TOL, tune="both")   # (default launch choices, then the tuned ones)


def test_full_config5_effnetv2m_realformer_vqa_asl_224():
    """BASELINE.json configs[4]: vqamed2019/train.py --loss=ASLSingleLabel, tf_efficientnetv2_m + RealFormer, VQA head
    (masked mean-pool) with 1552 answer classes, T 28 (the script's default), full depth and width, 224x224, batch 4"""
    run_case(O.make_args(cnn_encoder="tf_efficientnetv2_m", transformer_model="realformer", heads=8, dataset="VQA-Med",
                         vocab_size=1552, emb_vocab=30522, hidden_dropout_prob=0.0, emb_dropout_prob=0.0,
                         rf_dropout_prob=0.0), B=4, T=28, hw=224, kind="vqa", stat_tol=TOL, tune="both")


def test_full_config5_one_image_batch():
    """the smallest batch the path accepts, full depth and width: batch statistics over the pixels of a single 128x128
    image, squeeze-excite and ASL on one row"""
    run_case(O.make_args(cnn_encoder="tf_efficientnetv2_m", transformer_model="realformer", heads=8, dataset="VQA-Med",
                         vocab_size=1552, emb_vocab=30522, hidden_dropout_prob=0.0, emb_dropout_prob=0.0,
                         rf_dropout_prob=0.0), B=1, T=28, hw=128, kind="vqa", stat_tol=TOL)


@pytest.mark.parametrize("cfg", ["config2", "config3", "config5"])
def test_tuned_launch_choices_give_the_same_step(cfg):
    """bench.py / train.py call Model.tune(): every GEMM shape of the step then runs with the tile, split-K factor and
    (for products with few output tiles) the K split over workgroups + finishing launch the timed tuner picked -- other
    kernels than the untuned defaults the parity tests above go through.  Full width, 224x224 and the bench's batch, so
    that every GEMM shape of the full models occurs; two blocks per ResNet layer / a quarter of the EfficientNet
    repeats keep the chaotic amplification of fp32 reordering noise through 150 train-mode BatchNorms out of the
    comparison.  The tuned step must give the logits, loss and every gradient of the untuned one (same arithmetic,
    another summation order), and tuning itself must leave parameters, buffers and gradients untouched."""
    if cfg == "config2":
        args = O.make_args(resnet_layers=(2, 2, 2, 2), hidden_dropout_prob=0.0, emb_dropout_prob=0.0, rf_dropout_prob=0.0)
        B, T, kind = 16, 32, "mlm"
    elif cfg == "config3":
        args = O.make_args(cnn_encoder="tf_efficientnetv2_m", effnet_depth_div=4, transformer_model="realformer", heads=8,
                           supcon=True, hidden_dropout_prob=0.0, emb_dropout_prob=0.0, rf_dropout_prob=0.0)
        B, T, kind = 16, 32, "supcon"
    else:
        args = O.make_args(cnn_encoder="tf_efficientnetv2_m", effnet_depth_div=4, transformer_model="realformer", heads=8,
                           dataset="VQA-Med", vocab_size=1552, emb_vocab=30522, hidden_dropout_prob=0.0,
                           emb_dropout_prob=0.0, rf_dropout_prob=0.0)
        B, T, kind = 32, 28, "vqa"
    torch.manual_seed(3)
    hip = mmvqa_amd.Model(args)
    hip.to(dev()).train()
    if kind == "vqa":
        batch = synth.vqa_batch(B, T, 224, vocab=30522, n_classes=1552, seed=9, device=dev())
    else:
        batch = synth.roco_batch(B, T, 224, vocab=30522, seed=9, device=dev(), mlm_prob=0.3)
    img, ids, seg, mask, tgt = batch

    def step():
        hip.flat_grads.zero_()
        out = hip(img, ids, seg, mask)   # (`img` is rebound by the control below)
        if kind == "vqa":
            logits = out[0]
            loss = mmvqa_amd.asl_loss(logits, tgt)
        elif kind == "supcon":
            logits = out[0]
            loss = mmvqa_amd.mlm_loss(logits, tgt)[0] + mmvqa_amd.supcon_loss(mmvqa_amd.split_feat(out[1], B // 2))
        else:
            logits = out
            loss = mmvqa_amd.mlm_loss(logits, tgt)[0]
        loss.backward()
        torch.cuda.synchronize()
        return logits.detach().clone(), float(loss), hip.flat_grads.detach().clone()

    # control: the UNTUNED step on inputs moved by one unit in the last place.  Tuning changes tiles and K splits, i.e. the
    # order of every fp32 sum of the step; that perturbs every activation in its last bit, and train-mode BatchNorm at
    # random init amplifies last-bit perturbations on the way up and down the network.  How much is measured directly:
    # the per-tensor spread n0 of three runs whose image differs by +-1 ulp per pixel ("the same arithmetic on data that
    # differs as little as a reordered sum does").  Run-to-run noise of identical inputs is far smaller (the statistics
    # are accumulated in fp64) and would not be a fair yardstick.
    l0, loss0, g0 = step()
    img_exact = img
    ctl = []
    for k in range(3):
        gen = torch.Generator(device=dev()).manual_seed(17 + k)
        flip = (torch.rand(img_exact.shape, generator=gen, device=dev()) < 0.5).float() * 2 - 1
        img = img_exact * (1.0 + 6e-8 * flip)
        ctl.append(step())
    img = img_exact
    p_before, b_before = hip.flat_params.detach().clone(), hip._flat[1].clone()
    n = hip.tune(img, ids, seg, mask)
    assert n > 20, n
    assert torch.equal(hip.flat_params, p_before) and torch.equal(hip._flat[1], b_before)
    assert float(hip.flat_grads.abs().max()) == 0.0
    l1, loss1, g1 = step()
    ln0 = max(relerr(c[0], l0) for c in ctl)
    assert relerr(l1, l0) <= max(1e-4, 3 * ln0), f"logits tuned vs untuned {relerr(l1, l0):.2e} (one-ulp control {ln0:.2e})"
    assert abs(loss1 - loss0) <= 5e-4 * abs(loss0), (loss1, loss0)
    # per parameter tensor, relative to that tensor's largest gradient -- floored at 1e-3 of the largest gradient of the
    # model: a bias in front of a BatchNorm (projection-BN beta, proj_k.bias) has an exact gradient of zero and what
    # either run computes for it is rounding noise.  Bound: 4 x the tensor's own spread n0 under the one-ulp control
    # (floor 1e-4; measured on config 2: n0 = 4e-3 .. 1e-2 of a tensor's largest gradient, tuned-vs-untuned 1.3e-2 at most).  A wrong tile / split variant shows up as an O(1) error of the tensors it touches.
    floor = 1e-3 * float(g0.abs().max())
    bad, report = [], []
    for name, prm in hip.named_parameters():
        o, k = (prm.data_ptr() - hip.flat_params.data_ptr()) // 4, prm.numel()
        b = g0[o:o + k]
        scale = max(float(b.abs().max()), floor)
        n0 = max(float((c[2][o:o + k] - b).abs().max()) for c in ctl) / scale
        e = float((g1[o:o + k] - b).abs().max()) / scale
        report.append((e, n0, name))
        if e > max(4 * n0, 1e-4):
            bad.append(f"{name}: tuned-vs-untuned {e:.2e}, untuned under a 1-ulp input change {n0:.2e}")
    report.sort(reverse=True)
    print("largest tuned-vs-untuned differences (e, n0, tensor):", [(f"{e:.1e}", f"{n0:.1e}", nm) for e, n0, nm in report[:5]])
    assert not bad, "gradients move more under tuning than under a one-ulp change of the input: " + "; ".join(bad[:8])
    assert relerr(g1, g0) <= max(1e-3, 3 * max(relerr(c[2], g0) for c in ctl)), f"all gradients tuned vs untuned {relerr(g1, g0):.2e}"


@pytest.mark.parametrize("tag,tm,ds,supcon,cnn,relu", MODEL_CASES)
def test_golden_reference_replay(golden_dir, tag, tm, ds, supcon, cnn, relu):
    """inputs/outputs recorded from the REFERENCE's own Model.forward (tests/golden/make_golden.py), incl. its
    Timm_EFfNetV2 tap path at the shapes of BASELINE configs[2] (MLM), [3] (MLM + SupCon), [4] (VQA + ASL)"""
    g = dict(np.load(os.path.join(golden_dir, tag + ".npz"), allow_pickle=False))
    B, T, hw, V = [int(v) for v in g["dims"]]
    args = O.make_args(**model_case_args(tm, ds, supcon, cnn, relu, V, emb_dropout_prob=0.0, rf_dropout_prob=0.0,
                                         emb_vocab=V))
    torch.manual_seed(int(g["seed"]))
    orc = O.OracleModel(args)   # same seeded weights the reference ran with
    zero_dropout(orc)
    import copy
    orc64 = copy.deepcopy(orc).double().train()
    hip = mmvqa_amd.Model(args)
    hip.load_state_dict(orc.state_dict())
    hip.to(dev()).train()
    t = lambda k: torch.from_numpy(g[k]).to(dev())  # noqa: E731
    out = hip(t("img"), t("ids"), t("seg"), t("mask"))
    tgt = t("target")
    if ds == "roco":
        logits = out[0] if supcon else out
        loss = mmvqa_amd.mlm_loss(logits, tgt)[0]
        if supcon:
            assert relerr(out[1], torch.from_numpy(g["feat"])) <= TOL
            loss = loss + mmvqa_amd.supcon_loss(mmvqa_amd.split_feat(out[1], B // 2))
    else:
        logits = out[0]
        loss = mmvqa_amd.asl_loss(logits, tgt)
    e = relerr(logits, torch.from_numpy(g["logits"]))
    assert e <= TOL, f"logits vs reference {e:.2e}"
    assert abs(float(loss) - float(g["loss"])) <= TOL * abs(float(g["loss"]))
    loss.backward()
    # fp64 run of the oracle: how far the reference's own fp32 gradients are from exact arithmetic
    tc = lambda k: torch.from_numpy(g[k])  # noqa: E731
    o64 = orc64(tc("img").double(), tc("ids"), tc("seg"), tc("mask"))
    if ds == "roco":
        l64 = O.mlm_loss(o64[0] if supcon else o64, tc("target"))[0]
        if supcon:
            l64 = l64 + O.supcon_simclr(O.split_feat(o64[1], B // 2))
    else:
        l64 = O.asl_single_label(o64[0], tc("target"))
    l64.backward()
    p64 = dict(orc64.named_parameters())
    hp = dict(hip.named_parameters())

    def sub(t):
        return t.flatten()[:: max(1, t.numel() // 4096)][:4096] if t.numel() > 8192 else t

    for k in g:
        if not k.startswith("g_"):
            continue
        name = k[2:].replace("__", ".")
        ref32, truth = torch.from_numpy(g[k]), sub(p64[name].grad)
        noise = relerr(ref32, truth)
        e = relerr(sub(hp[name].grad), truth)
        assert e <= max(2 * TOL, 5 * noise), f"{name}: {e:.2e} (reference fp32 noise {noise:.2e})"
    hsd = hip.state_dict()
    for k in g:
        if k.startswith("b_"):
            assert relerr(hsd[k[2:].replace("__", ".")].double(), torch.from_numpy(g[k]).double()) <= 1e-4, k


def test_eval_mode_and_state_dict_roundtrip():
    args = mini_args()
    orc, hip = build_pair(args, seed=3)
    img, ids, seg, mask, tgt = synth.roco_batch(3, 12, 32, vocab=50, seed=9)
    orc.eval()
    hip.eval()
    with torch.no_grad():
        ref = orc(img, ids, seg, mask)
        out = hip(img.to(dev()), ids.to(dev()), seg.to(dev()), mask.to(dev()))
    assert relerr(out, ref) <= TOL
    sd = {k: v.cpu() for k, v in hip.state_dict().items()}
    hip2 = mmvqa_amd.Model(args)
    hip2.load_state_dict(sd)
    hip2.to(dev()).eval()
    with torch.no_grad():
        out2 = hip2(img.to(dev()), ids.to(dev()), seg.to(dev()), mask.to(dev()))
    assert relerr(out2, out) <= 1e-5   # float atomics in the tap mean make the sum order run-dependent


def test_rehead_surgery_and_fused_adam():
    """vqamed2019/train.py:125-149 (classifier[2] swap) then one optimizer step vs torch.optim.Adam"""
    args = mini_args(dataset="VQA-Med", vocab_size=50)
    orc, hip = build_pair(args, seed=4)
    torch.manual_seed(7)
    new_head = torch.nn.Linear(96, 17)
    orc.classifier[2] = torch.nn.Linear(96, 17)
    orc.classifier[2].load_state_dict(new_head.state_dict())
    hip.classifier[2] = new_head
    assert hip.state_dict()["classifier.2.weight"].shape == (17, 96)
    assert str(hip.flat_params.device).startswith("cuda")
    img, ids, seg, mask, tgt = synth.vqa_batch(3, 10, 32, vocab=50, n_classes=17, seed=2)
    lr = 1e-3
    opt_ref = torch.optim.Adam(orc.parameters(), lr=lr)
    opt = mmvqa_amd.FusedAdam(hip, lr=lr)
    orc.train()
    hip.train()
    before = {k: v.detach().clone() for k, v in orc.named_parameters()}
    opt_ref.zero_grad()
    O.asl_single_label(orc(img, ids, seg, mask)[0], tgt).backward()
    opt_ref.step()
    mmvqa_amd.asl_loss(hip(img.to(dev()), ids.to(dev()), seg.to(dev()), mask.to(dev()))[0], tgt.to(dev())).backward()
    opt.step()
    # Adam's first step moves every element by -lr*sign(g): elements whose gradient is ~0 may flip, the
    # rest must agree; never-used parameters (grad None in the reference) must not move at all.
    hp = dict(hip.named_parameters())
    for k, p in orc.named_parameters():
        d_ref = p.detach() - before[k]
        d_hip = hp[k].detach().cpu() - before[k]
        if p.grad is None:
            assert float(d_hip.abs().max()) == 0.0, k
            continue
        flipped = ((d_ref - d_hip).abs() > 0.5 * lr).float().mean().item()
        assert flipped <= 0.03, f"{k}: {flipped:.3f} of the elements moved differently"
    assert float(hip.flat_grads.abs().max()) == 0.0   # zero_grad folded into the Adam pass


def test_feat_may_be_dropped_before_backward():
    """supcon_utils.py:283-284 rebinds `feat` to split_feat(feat) before backward: the engine must not depend on the
    caller keeping the tensor it returned alive (its block may be reused by the loss temporaries)"""
    args = mini_args(transformer_model="realformer", supcon=True)
    _, hip = build_pair(args, seed=8)
    img, ids, seg, mask, tgt = (t.to(dev()) for t in synth.roco_batch(4, 11, 32, vocab=50, seed=6, mlm_prob=0.3))
    hip.train()
    logits, feat = hip(img, ids, seg, mask)
    loss = mmvqa_amd.mlm_loss(logits, tgt)[0] + mmvqa_amd.supcon_loss(mmvqa_amd.split_feat(feat, 2))
    loss.backward()
    torch.cuda.synchronize()
    want = hip.flat_grads.clone()
    hip.flat_grads.zero_()
    logits, feat = hip(img, ids, seg, mask)
    fptr, fshape = feat.data_ptr(), feat.shape
    feat = mmvqa_amd.split_feat(feat, 2)          # the only reference to the returned tensor is gone
    junk = [torch.full(fshape, float("nan"), device=dev()) for _ in range(8)]   # the caching allocator hands the freed block out again
    reused = any(j.data_ptr() == fptr for j in junk)   # (allocator policy: informative only; the check is the gradients below)
    loss = mmvqa_amd.mlm_loss(logits, tgt)[0] + mmvqa_amd.supcon_loss(feat)
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(hip.flat_grads).all(), f"NaN reached the gradients (returned block reused: {reused})"
    assert relerr(hip.flat_grads, want) <= 1e-5


def test_stock_torch_adam_steps_the_flat_buffer():
    """the reference's `optim.Adam(model.parameters(), lr)` (roco_train.py:90) works unchanged on the
    parameter views -- two steps give the same parameters as FusedAdam"""
    args = mini_args()
    _, a = build_pair(args, seed=9)
    _, b = build_pair(args, seed=9)
    batches = [tuple(t.to(dev()) for t in synth.roco_batch(3, 12, 32, vocab=50, seed=20 + i, mlm_prob=0.3)) for i in range(2)]
    lr = 1e-3
    opt_a = mmvqa_amd.FusedAdam(a, lr=lr)
    a.train(), b.train()
    opt_b = None
    for img, ids, seg, mask, tgt in batches:
        opt_a.zero_grad()
        mmvqa_amd.mlm_loss(a(img, ids, seg, mask), tgt)[0].backward()
        opt_a.step()
        if opt_b is None:
            opt_b = torch.optim.Adam(b.parameters(), lr=lr)
        opt_b.zero_grad()                       # set_to_none=True: .grad is re-attached by the next backward
        mmvqa_amd.mlm_loss(b(img, ids, seg, mask), tgt)[0].backward()
        opt_b.step()
    torch.cuda.synchronize()
    used = [n for n, p in b.named_parameters() if p.grad is not None]
    assert len(used) > 30 and not any(n.startswith("transformer.blocks.norm2") for n in used)
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    for n in pa:   # (elements whose gradient is ~0 are decided by the order of the float atomics: allow 1 %)
        if n.endswith("proj_k.bias"):   # gradient identically 0 in exact arithmetic (softmax shift invariance): pure noise
            continue
        off = ((pa[n].detach() - pb[n].detach()).abs() > 0.05 * lr).float().mean().item()
        assert off <= 0.01, f"{n}: {off:.3f} of the elements differ between FusedAdam and torch.optim.Adam after two steps"


@pytest.mark.parametrize("cnn", ["resnet152", "tf_efficientnetv2_m"])
def test_adam_beside_the_backward_pass(cnn):
    """FusedAdam.overlap_backward(): the update of a gradient range is enqueued from backward's announcement of it, on a
    stream of its own.  (1) The announcements of one backward partition the flat buffer, every range is updated exactly
    once and step() has nothing left; (2) driven by hand on FIXED gradients the ranged launches give bit-identical
    parameters / moments to the one-launch form, incl. a range nobody announced; (3) three training steps land where
    the one-launch optimizer lands (the float atomics of the weight gradients differ run to run, hence a tolerance)."""
    kw = dict(cnn_encoder=cnn, effnet_depth_div=8) if cnn != "resnet152" else {}
    args = mini_args(**kw)
    _, a = build_pair(args, seed=11)
    _, b = build_pair(args, seed=11)
    a.train(), b.train()
    lr = 1e-3
    opt_a, opt_b = mmvqa_amd.FusedAdam(a, lr=lr), mmvqa_amd.FusedAdam(b, lr=lr)
    n = b.flat_params.numel()
    # (2) by hand: same gradients, ranged vs one launch
    g = torch.randn(n, device=dev())
    a.flat_grads.copy_(g), b.flat_grads.copy_(g)
    opt_b.overlap_backward()
    ev = torch.cuda.Event()
    ev.record()
    cuts = [0, 4 * (n // 12), 4 * (n // 7), 4 * (n // 5), n - 4 * (n // 9), n]   # [cuts[3], cuts[4]) is never announced
    for lo, hi in ((cuts[4], cuts[5]), (cuts[0], cuts[1]), (cuts[2], cuts[3]), (cuts[1], cuts[2])):
        opt_b._on_ready(lo, hi, ev)
    opt_a.step(), opt_b.step()
    torch.cuda.synchronize()
    assert torch.equal(a.flat_params, b.flat_params) and torch.equal(opt_a.m, opt_b.m) and torch.equal(opt_a.v, opt_b.v)
    assert float(b.flat_grads.abs().max()) == 0.0 and opt_b._done == []
    # (1) + (3) in the loop
    seen = []
    early = opt_b._early
    opt_b._early = lambda lo, hi, **k: (seen.append((lo, hi)), early(lo, hi, **k))[1]
    batches = [tuple(t.to(dev()) for t in synth.roco_batch(3, 12, 32, vocab=50, seed=40 + i, mlm_prob=0.3)) for i in range(3)]
    for img, ids, seg, mask, tgt in batches:
        for m, o in ((a, opt_a), (b, opt_b)):
            o.zero_grad()
            mmvqa_amd.mlm_loss(m(img, ids, seg, mask), tgt)[0].backward()
            if o is opt_b:
                assert sorted(opt_b._done) == sorted(seen[-len(opt_b._done):]) and sum(h - l for l, h in opt_b._done) == n
                pos = 0
                for lo, hi in sorted(opt_b._done):
                    assert lo == pos, (lo, pos)
                    pos = hi
            o.step()
    torch.cuda.synchronize()
    assert len(seen) >= 3 * 3
    assert float(b.flat_grads.abs().max()) == 0.0
    off = ((a.flat_params - b.flat_params).abs() > 0.05 * lr).float().mean().item()
    assert off <= 0.01, f"{off:.4f} of the parameters differ between the two forms after three steps"


def test_dropout_training_mode():
    args = mini_args(hidden_dropout_prob=0.3, emb_dropout_prob=0.1)
    _, hip = build_pair(args, seed=5)
    img, ids, seg, mask, tgt = (t.to(dev()) for t in synth.roco_batch(4, 16, 32, vocab=50, seed=3))
    hip.train()
    hip.set_seed(11)
    a = hip(img, ids, seg, mask)
    la = mmvqa_amd.mlm_loss(a, tgt)[0]
    la.backward()
    ga = hip.flat_grads.clone()
    hip.set_seed(11)
    hip.flat_grads.zero_()
    b = hip(img, ids, seg, mask)
    mmvqa_amd.mlm_loss(b, tgt)[0].backward()
    hip.set_seed(12)
    c = hip(img, ids, seg, mask)
    assert torch.isfinite(a).all() and torch.isfinite(ga).all()
    assert relerr(b, a) < 1e-5                                   # same seed -> same masks
    assert relerr(hip.flat_grads, ga) < 1e-3                     # (atomics reorder sums slightly)
    assert relerr(c, a) > 1e-3                                   # different seed -> different masks
    # finite-difference check of the training-mode gradient along a random direction (same mask)
    hip.set_seed(11)
    d = torch.randn_like(hip.flat_params) * (hip.flat_params.abs() > 0)
    d = d / d.norm()
    eps = 1e-2
    with torch.no_grad():
        hip.flat_params.add_(d, alpha=eps)
        hip.set_seed(11)
        lp = float(mmvqa_amd.mlm_loss(hip(img, ids, seg, mask), tgt)[0])
        hip.flat_params.add_(d, alpha=-2 * eps)
        hip.set_seed(11)
        lm = float(mmvqa_amd.mlm_loss(hip(img, ids, seg, mask), tgt)[0])
        hip.flat_params.add_(d, alpha=eps)
    fd = (lp - lm) / (2 * eps)
    an = float((ga * d).sum())
    assert abs(fd - an) <= 0.1 * max(abs(an), 1e-3), (fd, an)


def test_grad_ready_ranges_partition_the_buffer():
    """data-parallel overlap: the ranges announced during backward tile [0, n_params) exactly once, and by the
    time a range is announced the current stream is ordered behind its gradients (values equal the final ones)"""
    for kw in (dict(resnet_layers=(2, 2, 14, 2), resnet_width=8), dict(cnn_encoder="tf_efficientnetv2_m", effnet_depth_div=3)):
        args = mini_args(**kw)
        _, hip = build_pair(args, seed=6)
        img, ids, seg, mask, tgt = (t.to(dev()) for t in synth.roco_batch(2, 12, 32, vocab=50, seed=4))
        seen, snaps = [], []

        def hook(lo, hi):
            seen.append((lo, hi))
            snaps.append((lo, hi, hip.flat_grads[lo:hi].clone()))   # stream-ordered copy at announcement time

        hip.set_grad_ready_hook(hook)
        hip.train()
        mmvqa_amd.mlm_loss(hip(img, ids, seg, mask), tgt)[0].backward()
        torch.cuda.synchronize()
        n = hip.flat_grads.numel()
        cover = sorted(seen)
        assert cover[0][0] == 0 and cover[-1][1] == n and len(cover) >= 4
        assert all(a[1] == b[0] for a, b in zip(cover[:-1], cover[1:])), cover
        for lo, hi, snap in snaps:
            assert torch.equal(snap, hip.flat_grads[lo:hi]), (lo, hi)
        hip.set_grad_ready_hook(None)


def test_grad_ready_event_orders_a_third_stream():
    """What RCCL's stream does with an announced range, without RCCL: a THIRD stream that waits only on the `ready`
    event of the announcement copies the range and then poisons it (NaN).  If the event really follows every writer of
    the range -- the engine stream's own kernels and the side stream's weight gradients joined before the
    announcement -- then (a) each copy equals the gradients of an undisturbed run and (b) the poison survives to the
    end of backward everywhere (a kernel that still wrote into an announced range would leave finite values).  gloo
    cannot show this: its CUDA path synchronises the producing stream on the host before it copies."""
    for kw in (dict(resnet_layers=(2, 2, 14, 2), resnet_width=8), dict(cnn_encoder="tf_efficientnetv2_m", effnet_depth_div=3)):
        args = mini_args(**kw)
        _, hip = build_pair(args, seed=6)
        img, ids, seg, mask, tgt = (t.to(dev()) for t in synth.roco_batch(2, 12, 32, vocab=50, seed=4))
        hip.train()
        mmvqa_amd.mlm_loss(hip(img, ids, seg, mask), tgt)[0].backward()
        torch.cuda.synchronize()
        want = hip.flat_grads.clone()
        hip.flat_grads.zero_()
        third = torch.cuda.Stream()
        snaps = []

        def hook(lo, hi, ready):
            with torch.cuda.stream(third):
                third.wait_event(ready)
                snaps.append((lo, hi, hip.flat_grads[lo:hi].clone()))
                hip.flat_grads[lo:hi].fill_(float("nan"))

        hip.set_grad_ready_hook(hook, with_event=True)
        mmvqa_amd.mlm_loss(hip(img, ids, seg, mask), tgt)[0].backward()
        torch.cuda.synchronize()
        hip.set_grad_ready_hook(None)
        assert len(snaps) >= 4
        assert bool(torch.isnan(hip.flat_grads).all()), "a kernel wrote into a gradient range after it was announced"
        scale = float(want.abs().max())
        for lo, hi, snap in snaps:
            assert bool(torch.isfinite(snap).all()), (lo, hi)
            err = float((snap - want[lo:hi]).abs().max()) / scale
            # (two runs differ by the order of their float atomics -- a few 1e-5 on this two-sample batch; a writer the
            # event does not cover would leave a whole contribution out)
            assert err <= 1e-3, f"range [{lo}, {hi}) read through the ready event differs from the finished gradients: {err:.2e}"
        hip.flat_grads.zero_()
