"""The N>1 path on CPU: two gloo ranks exercise the bucketed gradient all-reduce and the SupCon
feature all-gather; the same code runs over RCCL on the GPUs."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mmvqa_amd.ddp import GradReducer, comm_info, global_supcon_views, sync_replicas
    from oracle import mmbert_oracle as O
    try:
        # 1. bucketed all-reduce of a flat gradient buffer (buckets smaller than the buffer, ragged tail)
        torch.manual_seed(100 + rank)
        g = torch.randn(10007)
        mine = g.clone()
        red = GradReducer(g, bucket_mb=0.01)
        assert len(red.buckets) > 3 and red.buckets[0][1] == 10007 and red.buckets[-1][0] == 0
        red.allreduce()
        torch.manual_seed(100 + (1 - rank))
        other = torch.randn(10007)
        ok1 = torch.allclose(g, mine + other, atol=1e-6)
        # 2. partial range first (heads/encoder), then the rest: same result
        g2 = mine.clone()
        red2 = GradReducer(g2, bucket_mb=0.01)
        red2.start(lo=6000)
        red2.start(hi=6000)
        red2.finish()
        ok2 = torch.allclose(g2, mine + other, atol=1e-6)
        # 2b. the per-bucket callback an overlapped optimizer hangs on (optim.FusedAdam.overlap_backward): called once per
        # exchanged range with that range's work handle; the ranges partition the buffer; after work.wait() the range is reduced
        g3 = mine.clone()
        red3 = GradReducer(g3, bucket_mb=0.01)
        seen = []

        def after(lo, hi, work=None, stream=None):
            work.wait()
            seen.append((lo, hi, bool(torch.allclose(g3[lo:hi], (mine + other)[lo:hi], atol=1e-6))))
        red3.after_bucket = after
        assert red3.exchanges()
        red3.start(lo=6000)
        red3.start(hi=6000)
        red3.finish()
        pos = 0
        for lo, hi, good in sorted(seen):
            ok2 = ok2 and lo == pos and good
            pos = hi
        ok2 = ok2 and pos == 10007 and len(seen) >= len(red3.buckets)
        # 3. SupCon over the global view set: loss identical on both ranks, gradient = slice of the global one
        torch.manual_seed(7)
        full = torch.nn.functional.normalize(torch.randn(2 * world, 2, 16), dim=2)   # [N_global, 2 views, D]
        n = 2
        # rank r owns samples r*n .. r*n+n ; model output order per rank is view-major: [v1 of its n, v2 of its n]
        local = torch.cat([full[rank * n:(rank + 1) * n, 0], full[rank * n:(rank + 1) * n, 1]], 0).requires_grad_(True)
        feats = global_supcon_views(local, n)                       # the function train.py's supcon loop calls
        assert torch.equal(feats, full)
        loss = O.supcon_simclr(feats)
        loss.backward()
        ref_in = full.clone().requires_grad_(True)
        ref = O.supcon_simclr(ref_in)
        ref.backward()
        gref = torch.cat([ref_in.grad[rank * n:(rank + 1) * n, 0], ref_in.grad[rank * n:(rank + 1) * n, 1]], 0)
        ok3 = abs(float(loss) - float(ref)) < 1e-6 and torch.allclose(local.grad, gref * world, atol=1e-5)
        # 4. start-up: replicas that were built differently are made rank 0's, and the checksum agrees afterwards
        import types
        torch.manual_seed(500 + rank)
        fake = types.SimpleNamespace(_flat=[torch.randn(1003), torch.randn(77), torch.full((5,), rank, dtype=torch.long)])
        cs = sync_replicas(fake)
        torch.manual_seed(500)
        ok4 = torch.equal(fake._flat[0], torch.randn(1003)) and torch.equal(fake._flat[1], torch.randn(77)) and int(fake._flat[2].sum()) == 0
        info = comm_info(red)
        ok4 = ok4 and info["backend"] == "gloo" and info["world_size"] == world and info["buckets"] == len(red.buckets) and len(cs) == 4
        q.put((rank, ok1, ok2, ok3 and ok4))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok1, ok2, ok3 in res:
        assert ok1, f"rank {rank}: all-reduce"
        assert ok2, f"rank {rank}: ranged all-reduce"
        assert ok3, f"rank {rank}: feature all-gather / SupCon gradient"
