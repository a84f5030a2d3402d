"""Data-parallel layer (new in this build: the reference is single-process).

One process per GPU, full replica, per-GPU batch = the reference's batch (weak scaling); BatchNorm
statistics stay per-GPU (the reference has no SyncBN semantics to match).  Gradients live in ONE flat
fp32 buffer, so the exchange is a few large RCCL all-reduces over xGMI instead of ~930 small ones:
buckets are cut from the END of the buffer first (heads/encoder gradients are complete before the
backbone's), each issued asynchronously so RCCL's stream runs beside the remaining compute; the
1/world_size scale is folded into the fused Adam (no extra pass).  SupCon features use an all-gather
whose backward returns each rank its own slice.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class NativeComm:
    """RCCL communicator behind the C ABI of include/mmvqa_comm.h (libmmvqa_comm.so): the `mmvqa_allreduce_bucket` /
    `mmvqa_allgather` "thin wrappers over RCCL comms created by the Python launcher" of SURVEY.md 8(b).  Rank 0 draws the
    rendezvous id; the other ranks get its bytes over torch.distributed (any backend); one communicator per process on
    the current device.  Collectives are enqueued on the given (default: current) HIP stream and return at once."""

    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            import ctypes as C
            import os
            path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmmvqa_comm.so")
            if not os.path.exists(path):
                raise RuntimeError(f"{path} not found: build it with `make -C mm-vqa_amd/csrc` (needs RCCL)")
            L = C.CDLL(path)
            L.mmvqa_comm_last_error.restype = C.c_char_p
            P, LL = C.c_void_p, C.c_longlong
            for name, args in (("mmvqa_comm_unique_id", [P]), ("mmvqa_comm_create", [P, C.c_int, C.c_int, C.POINTER(P)]),
                               ("mmvqa_comm_destroy", [P]), ("mmvqa_comm_rank", [P]), ("mmvqa_comm_world", [P]),
                               ("mmvqa_comm_rccl_version", []), ("mmvqa_allreduce_bucket", [P, P, P, LL]),
                               ("mmvqa_allgather", [P, P, P, P, LL]), ("mmvqa_broadcast", [P, P, P, LL, C.c_int])):
                fn = getattr(L, name)
                fn.restype, fn.argtypes = C.c_int, args
            cls._lib = L
        return cls._lib

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("mmvqa_comm: " + self.lib().mmvqa_comm_last_error().decode())

    def __init__(self, rank=None, world_size=None):
        import ctypes as C
        L = self.lib()
        self.rank = (dist.get_rank() if dist.is_initialized() else 0) if rank is None else rank
        self.world = world() if world_size is None else world_size
        ident = C.create_string_buffer(128)
        if self.rank == 0:
            self._check(L.mmvqa_comm_unique_id(ident))
        if self.world > 1:
            box = [ident.raw]
            dist.broadcast_object_list(box, src=0)
            ident = C.create_string_buffer(box[0], 128)
        self._h = C.c_void_p()
        self._check(L.mmvqa_comm_create(ident, self.rank, self.world, C.byref(self._h)))

    @staticmethod
    def _stream(stream):
        return (stream or torch.cuda.current_stream()).cuda_stream

    def allreduce(self, t, stream=None):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        self._check(self.lib().mmvqa_allreduce_bucket(self._h, self._stream(stream), t.data_ptr(), t.numel()))

    def allgather(self, send, recv, stream=None):
        assert recv.numel() == send.numel() * self.world and send.is_contiguous() and recv.is_contiguous()
        self._check(self.lib().mmvqa_allgather(self._h, self._stream(stream), send.data_ptr(), recv.data_ptr(), send.numel()))

    def broadcast(self, t, root=0, stream=None):
        self._check(self.lib().mmvqa_broadcast(self._h, self._stream(stream), t.data_ptr(), t.numel(), root))

    def rccl_version(self):
        return self.lib().mmvqa_comm_rccl_version()

    def close(self):
        if getattr(self, "_h", None):
            self.lib().mmvqa_comm_destroy(self._h)
            self._h = None


class GradReducer:
    def __init__(self, flat_grads: torch.Tensor, bucket_mb: float = 64.0, native: "NativeComm | None" = None):
        """native: exchange the buckets through include/mmvqa_comm.h (`mmvqa_allreduce_bucket` on a communication stream
        of its own) instead of torch.distributed's all_reduce"""
        self.flat = flat_grads
        self.bucket_mb = float(bucket_mb)
        self.native = native
        self.comm_stream = torch.cuda.Stream() if native is not None else None
        self.native_pending = False
        n = flat_grads.numel()
        per = max(1, int(bucket_mb * (1 << 20) / 4))
        self.buckets = []
        hi = n
        while hi > 0:               # reverse order: tail of the buffer (heads, encoder) first
            lo = max(0, hi - per)
            self.buckets.append((lo, hi))
            hi = lo
        self.pending = []
        self.launched = 0
        self.after_bucket = None    # optim.FusedAdam.overlap_backward(): called per exchanged range with its work / stream

    def exchanges(self):
        """does start() exchange anything (else it only orders the stream)"""
        return world() > 1 or self.native is not None

    def start(self, lo=0, hi=None, ready=None):
        """launch async all-reduce (SUM) of every bucket inside [lo, hi).  This is the gradient-ready hook of
        Model.set_grad_ready_hook: called from inside backward, the collective is ordered behind the kernels
        that produced the range and runs beside the rest of the backward pass.
        `ready`: the event the engine's stream recorded AFTER it joined the side stream that carries the weight
        gradients of the range (Model.set_grad_ready_hook(..., with_event=True)).  The current stream waits on it
        before the collective is enqueued -- RCCL's stream then waits for the current stream, so the order
        side stream -> engine stream -> `ready` -> RCCL stream is explicit rather than implied by which stream
        happens to be current when the callback fires (gloo synchronises on the host and hides a missing edge)."""
        if ready is not None:
            torch.cuda.current_stream().wait_event(ready)
        if world() == 1 and self.native is None:
            return
        hi = self.flat.numel() if hi is None else hi
        if self.native is not None:
            # the communication stream is ordered behind the range's last writer (the ready event, or everything enqueued
            # on the current stream so far) and runs beside the rest of the backward pass
            ev = ready
            if ev is None:
                ev = torch.cuda.Event()
                ev.record()
            self.comm_stream.wait_event(ev)
            for a, b in self.buckets:
                a2, b2 = max(a, lo), min(b, hi)
                if a2 < b2:
                    self.native.allreduce(self.flat[a2:b2], stream=self.comm_stream)
                    if self.after_bucket is not None:
                        self.after_bucket(a2, b2, stream=self.comm_stream)
            self.native_pending = True
            self.launched += max(0, hi - lo)
            return
        for a, b in self.buckets:
            a2, b2 = max(a, lo), min(b, hi)
            if a2 < b2:
                w = dist.all_reduce(self.flat[a2:b2], op=dist.ReduceOp.SUM, async_op=True)
                self.pending.append(w)
                if self.after_bucket is not None:
                    self.after_bucket(a2, b2, work=w)
        self.launched += max(0, hi - lo)

    def finish(self):
        """all-reduce whatever backward did not announce (no hook installed), then wait for everything"""
        if (world() > 1 or self.native is not None) and self.launched == 0:
            self.start()
        for w in self.pending:
            w.wait()
        if self.native_pending:
            torch.cuda.current_stream().wait_stream(self.comm_stream)   # Adam reads the reduced gradients
            self.native_pending = False
        self.pending = []
        self.launched = 0

    def allreduce(self):
        self.finish()


def sync_replicas(model, src=0):
    """Make every rank's replica the one of rank `src` and prove it: broadcast the flat parameter buffer, the BatchNorm
    running statistics and the batch counters, then compare a checksum of all three across ranks (bit-equal or raise).
    Seeding alone (torch.manual_seed before Model(args)) gives identical replicas only while every rank builds the model
    with the same library versions and init order; the broadcast does not depend on that.  Returns the checksum."""
    flat_p, flat_b, flat_n = model._flat[0], model._flat[1], model._flat[2]
    w = world()
    if w > 1:
        for t in (flat_p, flat_b, flat_n):
            if t.numel():
                dist.broadcast(t, src=src)
    cs = torch.stack([flat_p.double().sum(), flat_p.double().abs().sum(), flat_b.double().sum(),
                      flat_n.double().sum() if flat_n.numel() else flat_p.new_zeros((), dtype=torch.float64)])
    if w > 1:
        got = [torch.empty_like(cs) for _ in range(w)]
        dist.all_gather(got, cs)
        for r, g in enumerate(got):
            if not torch.equal(g, got[0]):
                raise RuntimeError(f"replica of rank {r} differs from rank 0 after the broadcast: {g.tolist()} vs {got[0].tolist()}")
    return [float(v) for v in cs.tolist()]


def comm_info(reducer=None):
    """what the communicator saw, for the bench line / logs: backend, world size, RCCL version, bucket plan"""
    info = dict(backend=None, world_size=world(), rccl_version=None)
    if dist.is_available() and dist.is_initialized():
        info["backend"] = dist.get_backend()
        if info["backend"] == "nccl":
            try:
                info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as ex:     # the version query is informative only
                info["rccl_version"] = f"unavailable ({type(ex).__name__})"
    if reducer is not None:
        info["buckets"] = len(reducer.buckets)
        info["bucket_mb"] = reducer.bucket_mb
        info["gradient_mb"] = reducer.flat.numel() * 4 / (1 << 20)
    import os
    info["nccl_env"] = {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_"))}
    return info


class _AllGatherFeat(torch.autograd.Function):
    """features [n, D] per rank -> [world*n, D]; backward hands each rank the gradient of its own rows
    summed over ranks (every rank computes the same global SupCon loss)."""

    @staticmethod
    def forward(ctx, x):
        ws = world()
        ctx.n = x.shape[0]
        if ws == 1:
            return x
        out = [torch.empty_like(x) for _ in range(ws)]
        dist.all_gather(out, x.contiguous())
        return torch.cat(out, dim=0)

    @staticmethod
    def backward(ctx, g):
        ws = world()
        if ws == 1:
            return g
        g = g.contiguous()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        r = dist.get_rank()
        return g[r * ctx.n:(r + 1) * ctx.n]


def all_gather_features(x):
    return _AllGatherFeat.apply(x)


def global_supcon_views(feat, n):
    """feat [2n, D] of this rank (view-major: process_tensors, supcon_utils.py:253-256) -> [world*n, 2, D] over the
    global view set (split_feat layout, supcon_utils.py:259-261), so that every rank's samples are negatives of every
    other rank's (SURVEY 8(e) collective 2); differentiable: backward returns this rank's rows."""
    f = all_gather_features(feat)
    w = world()
    parts = f.view(w, 2, n, -1)
    return torch.cat([parts[:, 0].reshape(w * n, 1, -1), parts[:, 1].reshape(w * n, 1, -1)], 1).contiguous()
