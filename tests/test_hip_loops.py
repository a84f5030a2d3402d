"""Step-loop parity: mmvqa_amd.train's step functions (the callers the build supplies for
pretrain/roco_utils.py:214-247, models/SupConLoss/supcon_utils.py:270-294, vqamed2019/utils.py:633-666) against the
oracle restatement of those loops (oracle/loops_oracle.py) on identical weights and batches, dropout p = 0:
per-step loss, predictions (bit-exact), running accuracy, and the parameter change after two optimizer steps."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

import mmvqa_amd  # noqa: E402
from mmvqa_amd import synth, train  # noqa: E402
from mmvqa_amd.ddp import GradReducer  # noqa: E402
from oracle import loops_oracle as LO  # noqa: E402
from oracle import mmbert_oracle as O  # noqa: E402
from hip_helpers import dev  # noqa: E402
from test_hip_model import build_pair, mini_args  # noqa: E402

LR = 1e-3


def to_dev(batch):
    return tuple(tuple(x.to(dev()) for x in t) if isinstance(t, tuple) else t.to(dev()) for t in batch)


def check_param_deltas(orc, hip, before):
    """after the same optimizer steps both models moved the same way.  Adam's first steps move an element by about
    lr*sign(g): elements whose gradient is ~0 can flip with the summation order, the rest must agree."""
    hp = dict(hip.named_parameters())
    moved = 0
    for k, p in orc.named_parameters():
        d_ref = p.detach() - before[k]
        d_hip = hp[k].detach().cpu() - before[k]
        if p.grad is None:
            assert float(d_hip.abs().max()) == 0.0, f"{k}: the reference never updates it"
            continue
        if k.endswith("proj_k.bias"):     # gradient == 0 in exact arithmetic (softmax shift invariance): noise
            continue
        moved += 1
        off = ((d_ref - d_hip).abs() > 0.5 * LR).float().mean().item()
        assert off <= 0.03, f"{k}: {off:.3f} of the elements moved differently after two steps"
    assert moved > 30


def test_mlm_loop_matches_oracle_loop():
    args = mini_args()
    orc, hip = build_pair(args, seed=21)
    before = {k: v.detach().clone() for k, v in orc.named_parameters()}
    loader = [synth.roco_batch(3, 12, 32, vocab=50, seed=30 + i, mlm_prob=0.4) for i in range(2)]
    opt_ref = torch.optim.Adam(orc.parameters(), lr=LR)                 # roco_train.py:90
    ref_loss, ref_acc, ref_losses, ref_preds = LO.mlm_train_one_epoch(loader, orc, torch.nn.NLLLoss(), opt_ref)
    hip.train()
    opt = mmvqa_amd.FusedAdam(hip, lr=LR)
    red = GradReducer(hip.flat_grads)
    nm = nc = 0.0
    for i, b in enumerate(loader):
        b = to_dev(b)
        loss, pred, stats = train.mlm_step(hip, opt, red, 1, b)
        assert abs(float(loss) - float(ref_losses[i])) <= 1e-3 * abs(float(ref_losses[i])), (i, float(loss), float(ref_losses[i]))
        assert torch.equal(pred[b[4] > 0].cpu(), ref_preds[i]), f"step {i}: masked-position predictions differ"
        s = stats.tolist()
        assert s[1] == float((b[4] > 0).sum())
        nm, nc = nm + s[1], nc + s[2]
    assert abs(100.0 * nc / nm - ref_acc) < 1e-9
    check_param_deltas(orc, hip, before)


def test_supcon_loop_matches_oracle_loop():
    args = mini_args(transformer_model="realformer", supcon=True)
    orc, hip = build_pair(args, seed=22)
    before = {k: v.detach().clone() for k, v in orc.named_parameters()}
    loader = []
    for i in range(2):
        a = synth.roco_batch(3, 12, 32, vocab=50, seed=40 + 2 * i, mlm_prob=0.4)
        b = synth.roco_batch(3, 12, 32, vocab=50, seed=41 + 2 * i, mlm_prob=0.4)
        loader.append(((a[0], b[0]), a[1], b[1], a[2], a[3], a[4], b[4]))   # ROCO_SupCon item layout (supcon_utils.py:270)
    opt_ref = torch.optim.Adam(orc.parameters(), lr=LR)
    _, ref_acc, ref_losses, ref_preds = LO.supcon_train_one_epoch(loader, orc, torch.nn.NLLLoss(), O.supcon_simclr, opt_ref)
    hip.train()
    opt = mmvqa_amd.FusedAdam(hip, lr=LR)
    red = GradReducer(hip.flat_grads)
    for i, item in enumerate(loader):
        batch = train.process_tensors(*to_dev(item))
        ref_batch = LO.process_tensors(*item)
        assert all(torch.equal(x.cpu(), y) for x, y in zip(batch, ref_batch))
        loss, pred, stats = train.supcon_step(hip, opt, red, 1, batch)
        assert abs(float(loss) - float(ref_losses[i])) <= 1e-3 * abs(float(ref_losses[i])), (i, float(loss), float(ref_losses[i]))
        assert torch.equal(pred[batch[4] > 0].cpu(), ref_preds[i])
    check_param_deltas(orc, hip, before)


@pytest.mark.parametrize("loss_name,clip", [("ASLSingleLabel", False), ("CrossEntropyLoss", False), ("ASLSingleLabel", True)])
def test_vqa_loop_matches_oracle_loop(loss_name, clip):
    args = mini_args(transformer_model="realformer", dataset="VQA-Med", vocab_size=23)
    orc, hip = build_pair(args, seed=23)
    before = {k: v.detach().clone() for k, v in orc.named_parameters()}
    loader = [synth.vqa_batch(4, 10, 32, vocab=50, n_classes=23, seed=50 + i) for i in range(2)]
    crit_ref = O.asl_single_label if loss_name == "ASLSingleLabel" else torch.nn.CrossEntropyLoss()
    crit = mmvqa_amd.asl_loss if loss_name == "ASLSingleLabel" else (lambda lg, t: mmvqa_amd.mlm_loss(lg, t)[0])
    opt_ref = torch.optim.Adam(orc.parameters(), lr=LR)                 # vqamed2019/train.py:160
    _, ref_acc, ref_losses, ref_preds = LO.vqa_train_one_epoch(loader, orc, opt_ref, crit_ref, clip=clip)
    hip.train()
    opt = mmvqa_amd.FusedAdam(hip, lr=LR)
    red = GradReducer(hip.flat_grads)
    for i, b in enumerate(loader):
        loss, pred = train.vqa_step(hip, opt, red, 1, to_dev(b), crit, clip=clip)
        assert abs(float(loss) - float(ref_losses[i])) <= 1e-3 * abs(float(ref_losses[i])), (i, float(loss), float(ref_losses[i]))
        assert torch.equal(pred.cpu(), ref_preds[i])
    check_param_deltas(orc, hip, before)


# ----------------------------------------------------------------------------- replay of the REFERENCE's own loops
def _fixture_model(golden_dir, tag, tm, ds, supcon):
    import os
    import numpy as np
    from test_oracle_golden import model_case_args
    g = dict(np.load(os.path.join(golden_dir, tag + ".npz"), allow_pickle=False))
    B, T, hw, V = [int(v) for v in g["dims"]]
    args = O.make_args(**model_case_args(tm, ds, supcon, "resnet152", False, V, emb_dropout_prob=0.0, rf_dropout_prob=0.0,
                                         emb_vocab=V))
    torch.manual_seed(int(g["seed"]))
    orc = O.OracleModel(args)                       # the seeded weights the reference loop started from
    hip = mmvqa_amd.Model(args)
    hip.load_state_dict(orc.state_dict())
    hip.to(dev()).train()
    return g, hip


def _check_after(hip, g, lr):
    hp = dict(hip.named_parameters())
    n = 0
    for k in g:
        if not k.startswith("p_"):
            continue
        name = k[2:].replace("__", ".")
        got = hp[name].detach().flatten().cpu()
        if got.numel() > 4096:
            got = got[:: max(1, got.numel() // 4096)][:4096]
        off = ((got - torch.from_numpy(g[k])).abs() > 0.5 * lr).float().mean().item()
        assert off <= 0.03, f"{name}: {off:.3f} of the sampled elements differ from the reference after two steps"
        n += 1
    assert n >= 14
    sd = hip.state_dict()
    assert int(sd["transformer.trans.model.bn1.num_batches_tracked"]) == int(g["b_transformer__trans__model__bn1__num_batches_tracked"])
    rm = sd["transformer.trans.model.bn1.running_mean"].cpu()
    ref = torch.from_numpy(g["b_transformer__trans__model__bn1__running_mean"])
    assert float((rm - ref).abs().max()) <= 1e-3 * float(ref.abs().max())


def test_reference_loop_fixtures_replay(golden_dir):
    """fixtures recorded from the reference's OWN train_one_epoch functions (tests/golden/make_golden_loops.py):
    train.py's step functions reproduce the per-step losses and the parameters after two Adam steps"""
    tg = lambda g, k: torch.from_numpy(g[k]).to(dev())   # noqa: E731
    # MLM (pretrain/roco_utils.py:207-290)
    g, hip = _fixture_model(golden_dir, "loop_mlm", "transformer", "roco", False)
    lr = float(g["lr"])
    opt, red = mmvqa_amd.FusedAdam(hip, lr=lr), GradReducer(hip.flat_grads)
    nm = nc = 0.0
    for i in range(2):
        b = tuple(tg(g, f"{n}{i}") for n in ("img", "ids", "seg", "mask", "tgt"))
        loss, pred, stats = train.mlm_step(hip, opt, red, 1, b)
        assert abs(float(loss) - float(g["losses"][i])) <= 1e-3 * abs(float(g["losses"][i])), (i, float(loss))
        s = stats.tolist()
        nm, nc = nm + s[1], nc + s[2]
    assert abs(100.0 * nc / nm - float(g["total_acc"])) < 1e-9
    _check_after(hip, g, lr)
    # MLM + SupCon (models/SupConLoss/supcon_utils.py:263-323)
    g, hip = _fixture_model(golden_dir, "loop_supcon", "realformer", "roco", True)
    opt, red = mmvqa_amd.FusedAdam(hip, lr=lr), GradReducer(hip.flat_grads)
    for i in range(2):
        batch = train.process_tensors((tg(g, f"img_a{i}"), tg(g, f"img_b{i}")), tg(g, f"ids_a{i}"), tg(g, f"ids_b{i}"),
                                      tg(g, f"seg{i}"), tg(g, f"mask{i}"), tg(g, f"tgt_a{i}"), tg(g, f"tgt_b{i}"))
        loss, _, _ = train.supcon_step(hip, opt, red, 1, batch)
        want = float(g["losses_mlm"][i] + g["losses_supcon"][i])
        assert abs(float(loss) - want) <= 1e-3 * abs(want), (i, float(loss), want)
    _check_after(hip, g, lr)
    # VQA-Med + ASL (vqamed2019/utils.py:625-688)
    g, hip = _fixture_model(golden_dir, "loop_vqa", "realformer", "VQA-Med", False)
    opt, red = mmvqa_amd.FusedAdam(hip, lr=lr), GradReducer(hip.flat_grads)
    preds = []
    for i in range(2):
        b = tuple(tg(g, f"{n}{i}") for n in ("img", "ids", "seg", "mask", "tgt"))
        loss, pred = train.vqa_step(hip, opt, red, 1, b, mmvqa_amd.asl_loss)
        assert abs(float(loss) - float(g["losses"][i])) <= 1e-3 * abs(float(g["losses"][i])), (i, float(loss))
        preds.append(pred.cpu())
    assert torch.equal(torch.cat(preds), torch.from_numpy(g["preds"]))
    _check_after(hip, g, lr)
