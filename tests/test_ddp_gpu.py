"""The data-parallel step on the GPU with two processes: the engine's gradient-ready callback starts bucketed
all-reduces while the backward is still running (mmvqa_amd.ddp.GradReducer), and the reduced buffer equals the sum of
the per-rank gradients computed one after the other in a single process.  Both ranks share cuda:0 and talk over gloo
(one-GPU box; on the 8-GPU node the same code runs over RCCL, one rank per GPU)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mmvqa_amd
        from mmvqa_amd import synth
        from mmvqa_amd.ddp import GradReducer
        from oracle import mmbert_oracle as O
        dev = torch.device("cuda", 0)
        args = O.make_args(resnet_layers=(2, 2, 2, 2), resnet_width=16, hidden_size=96, n_layers=2, heads=12, vocab_size=64,
                           emb_vocab=64, bert_max_pos=32, hidden_dropout_prob=0.0, emb_dropout_prob=0.0, rf_dropout_prob=0.0)
        torch.manual_seed(0)                       # identical replicas
        model = mmvqa_amd.Model(args).to(dev).train()
        batches = [synth.roco_batch(3, 16, 64, vocab=64, seed=40 + r, device=dev, mlm_prob=0.4) for r in range(world)]

        def fwd_bwd(b):
            img, ids, seg, mask, tgt = b
            loss = mmvqa_amd.mlm_loss(model(img, ids, seg, mask), tgt)[0]
            loss.backward()

        red = GradReducer(model.flat_grads, bucket_mb=0.02)
        calls = []
        model.set_grad_ready_hook(lambda lo, hi: (calls.append((lo, hi)), red.start(lo, hi)))
        fwd_bwd(batches[rank])
        n_started = len(red.pending)
        red.allreduce()
        torch.cuda.synchronize()
        got = model.flat_grads.clone()
        # reference: every rank's batch through the same replica, one after the other, no hook
        model.set_grad_ready_hook(None)
        ref = torch.zeros_like(got)
        for r in range(world):
            model.flat_grads.zero_()
            fwd_bwd(batches[r])
            torch.cuda.synchronize()
            ref += model.flat_grads
        err = float((got - ref).abs().max() / ref.abs().max())
        covered = sum(hi - lo for lo, hi in calls)
        q.put((rank, err, len(calls), n_started, covered == got.numel()))
    finally:
        dist.destroy_process_group()


def test_two_rank_overlapped_allreduce_on_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, n_calls, n_started, covered in res:
        assert n_calls >= 3 and n_started >= n_calls, f"rank {rank}: hook calls {n_calls}, all-reduces started during backward {n_started}"
        assert covered, f"rank {rank}: announced ranges do not cover the buffer"
        assert err <= 1e-5, f"rank {rank}: reduced gradients differ from the sum of the per-rank gradients: {err:.2e}"


def _supcon_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mmvqa_amd
        from mmvqa_amd.ddp import global_supcon_views
        from oracle import mmbert_oracle as O
        dev = torch.device("cuda", 0)
        n, D = 64, 128                                  # 2n*world = 256 rows = BASELINE configs[3] on 8 GPUs
        torch.manual_seed(7)
        full = torch.nn.functional.normalize(torch.randn(n * world, 2, D), dim=2)       # [N_global, 2 views, D]
        mine = full[rank * n:(rank + 1) * n]
        local = torch.cat([mine[:, 0], mine[:, 1]], 0).to(dev).requires_grad_(True)    # model output order
        feats = global_supcon_views(local, n)           # all-gather (differentiable) + split_feat layout
        assert feats.shape == (n * world, 2, D)
        loss = mmvqa_amd.supcon_loss(feats)             # HIP kernel on the gathered set
        loss.backward()
        torch.cuda.synchronize()
        ref_in = full.clone().requires_grad_(True)
        ref = O.supcon_simclr(ref_in)
        ref.backward()
        g = ref_in.grad[rank * n:(rank + 1) * n]
        gref = torch.cat([g[:, 0], g[:, 1]], 0) * world   # every rank back-propagates the same global loss
        err = float((local.grad.cpu() - gref).abs().max() / gref.abs().max())
        q.put((rank, abs(float(loss) - float(ref)) / abs(float(ref)), err))
    finally:
        dist.destroy_process_group()


def test_two_rank_gathered_supcon_on_gpu():
    """all-gather of the [2n, 128] features -> HIP SupCon over 2n*world = 256 rows -> each rank gets the gradient of
    its own rows of the single-process global loss"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_supcon_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, lerr, gerr in res:
        assert lerr <= 2e-5, f"rank {rank}: global SupCon loss differs from the oracle: {lerr:.2e}"
        assert gerr <= 1e-4, f"rank {rank}: feature gradient differs from the slice of the global one: {gerr:.2e}"


def test_native_rccl_wrappers_on_a_one_rank_communicator():
    """include/mmvqa_comm.h on the one GPU of the test box: a world-1 RCCL communicator (two ranks cannot share a device
    under RCCL), all-reduce = identity, all-gather = copy, broadcast = no-op, all stream-ordered; and the GradReducer
    route that bench.py --native-comm takes (buckets on the communication stream, joined before the optimizer)"""
    import torch
    from mmvqa_amd.ddp import GradReducer, NativeComm
    dev = torch.device("cuda:0")
    c = NativeComm(rank=0, world_size=1)
    assert c.rccl_version() > 20000
    g = torch.randn(100003, device=dev)
    want = g.clone()
    c.allreduce(g)
    out = torch.zeros(4096, device=dev)
    c.allgather(want[:4096].contiguous(), out)
    c.broadcast(g)
    torch.cuda.synchronize()
    assert torch.equal(g, want) and torch.equal(out, want[:4096])
    red = GradReducer(g, bucket_mb=0.05, native=c)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g.mul_(2.0)                       # the "backward" producing the range on another stream
        ev = torch.cuda.Event()
        ev.record(s)
    red.start(50000, None, ready=ev)      # tail first, ordered behind its writer by the event
    red.start(0, 50000, ready=ev)
    red.finish()
    torch.cuda.synchronize()
    assert torch.equal(g, want * 2.0)
    c.close()
