"""Host-side mirror of the reference's ``Model(args)`` protocol over the HIP engine.

Reference interface mirrored here:
  * ``Model(args)``                      -- models/mmbert.py:129-148
  * ``model(img, ids, seg, mask)``       -- models/mmbert.py:150-167 (Tensor | (logits, feat) | (logits, 0, 0))
  * ``.to() .train() .eval() .parameters() .state_dict() .load_state_dict()``
  * ``model.classifier[2] = nn.Linear(hidden, num_classes)`` -- vqamed2019/train.py:137
Parameter names and logical shapes equal the reference's state_dict; storage is ONE flat fp32
buffer per kind (params / grads / BN buffers) owned by PyTorch, whose layout the C++ engine
defines (mmvqa_engine_tensor_info).  All arithmetic runs in libmmvqa_hip.so; there is no
PyTorch fallback -- forward() on a non-GPU tensor or without the library raises.
"""
from __future__ import annotations

import ctypes as C
import math

import torch
import torch.nn as nn

from . import _lib as L

# parameters the reference never uses => their .grad stays None
_NEVER_USED = ("transformer.blocks.norm2.", "transformer.trans.model.fc.")


def desc_from_args(args, feat_dim=128, n_classes=None) -> L.ModelDesc:
    """Translate the argparse Namespace the reference passes to Model(args)."""
    d = L.ModelDesc()
    enc = args.cnn_encoder
    if "resnet" in enc:
        d.cnn = 0
    elif "efficientnetv2" in enc:
        d.cnn = 1
    else:
        raise NotImplementedError(enc)  # models/image_encoding.py:41
    layers = getattr(args, "resnet_layers", (3, 8, 36, 3))
    for i in range(4):
        d.resnet_layers[i] = int(layers[i])
    d.resnet_width = int(getattr(args, "resnet_width", 64))
    d.effnet_depth_div = int(getattr(args, "effnet_depth_div", 1))
    tm = args.transformer_model
    if "feedback-transformer" in tm:
        raise NotImplementedError("feedback-transformer is outside the hot path (SURVEY.md section 2, row 9)")
    elif "realformer" in tm:
        d.encoder = 1
    elif "transformer" in tm:
        d.encoder = 0
    else:
        raise NotImplementedError(tm)  # models/mmbert.py:42
    d.hidden = int(args.hidden_size)
    d.heads = int(getattr(args, "heads", 12))
    d.n_layers = int(args.n_layers)
    d.emb_vocab = int(getattr(args, "emb_vocab", 30522))
    d.max_pos = int(getattr(args, "bert_max_pos", 512))
    d.type_vocab = 2
    d.num_vis = int(args.num_vis)
    if args.dataset == "roco":
        if getattr(args, "task", "MLM") != "MLM":
            raise NotImplementedError("only task='MLM' is on the hot path")
        d.head_kind = 0
    elif args.dataset == "VQA-Med":
        d.head_kind = 1
    else:
        raise NotImplementedError(args.dataset)
    d.n_classes = int(n_classes if n_classes is not None else args.vocab_size)
    d.supcon = 1 if (getattr(args, "supcon", False) and d.head_kind == 0) else 0
    d.feat_dim = int(feat_dim)
    d.use_relu = 1 if getattr(args, "use_relu", False) else 0
    d.p_drop = float(getattr(args, "hidden_dropout_prob", 0.3))
    d.p_emb_drop = float(getattr(args, "emb_dropout_prob", 0.1))
    d.p_rf_drop = float(getattr(args, "rf_dropout_prob", 0.1))
    return d


class _Node(nn.Module):
    """plain container so that dotted state_dict names resolve to a module tree"""


class _HeadSeq(_Node):
    """``model.classifier``: indexable like nn.Sequential; assigning item 2 re-heads the model
    (vqamed2019/train.py:137,141,149: ``model.classifier[2] = nn.Linear(hidden, num_classes)``)."""

    def __init__(self, owner):
        super().__init__()
        object.__setattr__(self, "_owner", owner)

    def __getitem__(self, i):
        return getattr(self, str(i))

    def __setitem__(self, i, module):
        if int(i) != 2 or not isinstance(module, nn.Linear):
            raise NotImplementedError("only classifier[2] = nn.Linear(...) is supported")
        self._owner._rehead(module)

    def __len__(self):
        return 3


class _ModelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, anchor, img, ids, seg, mask):
        ctx.model = model
        return model._engine_forward(img, ids, seg, mask)

    @staticmethod
    def backward(ctx, *grads):
        ctx.model._engine_backward(*grads)
        return (None,) * 6


class Model(nn.Module):
    def __init__(self, args, feat_dim=128, device=None):
        super().__init__()
        self._args = args
        self._feat_dim = feat_dim
        self.task = getattr(args, "task", "MLM")
        self.dataset = args.dataset
        self.supcon = bool(getattr(args, "supcon", False))
        self._handle = None
        self._plan_key = None
        self._ws = None
        self._seed_ctr = 0
        self._fwd_state = None
        dev = torch.device(device) if device is not None else torch.device("cpu")
        self._build(desc_from_args(args, feat_dim), dev, init=True)

    # ------------------------------------------------------------------ construction
    def _build(self, desc, dev, init, old_state=None):
        lib = L.lib()
        if self._handle is not None:
            lib.mmvqa_engine_destroy(self._handle)
            self._handle = None
        h = C.c_void_p()
        L.check(lib.mmvqa_engine_create(C.byref(desc), C.byref(h)))
        self._handle = h
        self._desc = desc
        self._plan_key = None
        n_p = lib.mmvqa_engine_param_floats(h)
        n_b = lib.mmvqa_engine_buf_floats(h)
        n_n = max(1, lib.mmvqa_engine_nbt_count(h))
        self._flat = {
            0: torch.zeros(n_p, dtype=torch.float32, device=dev),
            1: torch.zeros(max(4, n_b), dtype=torch.float32, device=dev),
            2: torch.zeros(n_n, dtype=torch.int64, device=dev),
        }
        self._flat_grad = torch.zeros(n_p, dtype=torch.float32, device=dev)
        self._anchor = torch.zeros(1, dtype=torch.float32, device=dev, requires_grad=True)
        # drop the old module tree
        for name in list(self._modules.keys()):
            del self._modules[name]
        self._table = []
        name_buf = C.create_string_buffer(256)
        kind, ndim, cl = C.c_int(), C.c_int(), C.c_int()
        shape = (C.c_longlong * 4)()
        off = C.c_longlong()
        for i in range(lib.mmvqa_engine_num_tensors(h)):
            L.check(lib.mmvqa_engine_tensor_info(h, i, name_buf, 256, C.byref(kind), C.byref(ndim),
                                                 C.byref(shape), C.byref(off), C.byref(cl)))
            name = name_buf.value.decode()
            shp = tuple(int(shape[k]) for k in range(ndim.value))
            self._table.append((name, kind.value, shp, int(off.value), bool(cl.value)))
        self._params_by_name = {}
        for name, kind, shp, off, cl in self._table:
            view = self._view(self._flat[kind], shp, off, cl)
            parts = name.split(".")
            mod = self
            for j, p in enumerate(parts[:-1]):
                if p not in mod._modules:
                    child = _HeadSeq(self) if (j == 0 and p == "classifier") else _Node()
                    mod.add_module(p, child)
                mod = mod._modules[p]
            if kind == 0:
                prm = nn.Parameter(view, requires_grad=True)
                mod.register_parameter(parts[-1], prm)
                self._params_by_name[name] = prm
            else:
                mod.register_buffer(parts[-1], view)
        if init:
            self._init_weights()
        if old_state is not None:
            own = self.state_dict()
            keep = {k: v for k, v in old_state.items() if k in own and own[k].shape == v.shape}
            self.load_state_dict(keep, strict=False)

    @staticmethod
    def _view(flat, shp, off, cl):
        n = 1
        for s in shp:
            n *= s
        v = flat[off:off + n]
        if len(shp) == 0:
            return v.view(())
        if cl and len(shp) == 4:
            return v.view(shp[0], shp[2], shp[3], shp[1]).permute(0, 3, 1, 2)
        return v.view(*shp)

    @torch.no_grad()
    def _init_weights(self):
        """Seedable random init (no network fetches -- SURVEY.md 8(b) 'Construction'): torch's default
        per-module inits; HF's N(0, 0.02) for the embedding tables; torchvision's kaiming_normal
        fan_out for the backbone convs."""
        shapes = {n: shp for n, _, shp, _, _ in self._table}
        for name, kind, shp, off, cl in self._table:
            t = self._view(self._flat[kind], shp, off, cl)
            if kind == 2:
                t.zero_()
            elif kind == 1:
                t.fill_(1.0 if name.endswith("running_var") else 0.0)
            elif name.endswith("embeddings.weight"):
                t.normal_(0.0, 0.02)
                if "word_embeddings" in name:
                    t[0].zero_()  # padding_idx row
            elif len(shp) == 4:
                if ".trans.model." in name:
                    nn.init.kaiming_normal_(t, mode="fan_out", nonlinearity="relu")
                else:
                    nn.init.kaiming_uniform_(t, a=math.sqrt(5))
            elif len(shp) == 2:
                nn.init.kaiming_uniform_(t, a=math.sqrt(5))
            elif name.endswith(".bias") and len(shapes.get(name[:-5] + ".weight", ())) == 2:
                bound = 1.0 / math.sqrt(shapes[name[:-5] + ".weight"][1])
                t.uniform_(-bound, bound)
            elif name.endswith(".weight"):
                t.fill_(1.0)  # LayerNorm / BatchNorm gamma
            else:
                t.zero_()     # LayerNorm / BatchNorm beta

    def _rehead(self, new_linear: nn.Linear):
        """classifier[2] surgery: rebuild the engine with the new class count, keep every other
        weight, take classifier.2.* from the given nn.Linear."""
        old = {k: v.detach().clone() for k, v in self.state_dict().items() if not k.startswith("classifier.2.")}
        dev = self._flat[0].device
        desc = desc_from_args(self._args, self._feat_dim, n_classes=new_linear.out_features)
        self._build(desc, dev, init=True, old_state=old)
        with torch.no_grad():
            self._params_by_name["classifier.2.weight"].copy_(new_linear.weight)
            if new_linear.bias is not None:
                self._params_by_name["classifier.2.bias"].copy_(new_linear.bias)

    # ------------------------------------------------------------------ nn.Module protocol
    def _apply(self, fn, recurse=True):
        """.to()/.cuda()/.cpu(): move the flat buffers and re-point every parameter at its view."""
        probe = fn(torch.zeros(1, dtype=torch.float32, device=self._flat[0].device))
        if probe.dtype != torch.float32:
            raise NotImplementedError("the MMBERT hot path is fp32 (SURVEY.md section 8)")
        for k in (0, 1):
            self._flat[k] = fn(self._flat[k])
        self._flat[2] = self._flat[2].to(self._flat[0].device)
        self._flat_grad = fn(self._flat_grad)
        self._anchor = torch.zeros(1, dtype=torch.float32, device=self._flat[0].device, requires_grad=True)
        self._plan_key = None
        self._ws = None
        for name, kind, shp, off, cl in self._table:
            view = self._view(self._flat[kind], shp, off, cl)
            parts = name.split(".")
            mod = self
            for p in parts[:-1]:
                mod = mod._modules[p]
            if kind == 0:
                prm = mod._parameters[parts[-1]]
                had_grad = prm.grad is not None
                prm.data = view
                prm.grad = self._view(self._flat_grad, shp, off, cl) if had_grad else None
            else:
                mod._buffers[parts[-1]] = view
        return self

    def __del__(self):
        try:
            if self._handle is not None:
                L.lib().mmvqa_engine_destroy(self._handle)
        except Exception:
            pass

    # ------------------------------------------------------------------ flat views for the fused optimizer / DDP
    @property
    def flat_params(self):
        return self._flat[0]

    @property
    def flat_grads(self):
        return self._flat_grad

    def attach_grads(self):
        """make every used parameter's .grad a view of the flat gradient buffer"""
        for name, kind, shp, off, cl in self._table:
            if kind == 0 and not name.startswith(_NEVER_USED):
                p = self._params_by_name[name]
                if p.grad is None:
                    p.grad = self._view(self._flat_grad, shp, off, cl)

    # ------------------------------------------------------------------ engine calls
    def _ensure_plan(self, B, T, IH, IW):
        key = (B, T, IH, IW)
        lib = L.lib()
        if self._plan_key != key:
            nbytes = lib.mmvqa_engine_plan(self._handle, B, T, IH, IW)
            if nbytes == 0:
                L.check(-1)
            if self._ws is None or self._ws.numel() < nbytes:
                self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self._flat[0].device)
            L.check(lib.mmvqa_engine_bind(self._handle, L.ptr(self._flat[0]), L.ptr(self._flat_grad),
                                          L.ptr(self._flat[1]), L.ptr(self._flat[2]), L.ptr(self._ws),
                                          self._ws.numel()))
            self._plan_key = key

    def set_seed(self, seed: int):
        """dropout stream for training-mode forwards (counter-based RNG in the kernels)"""
        self._seed_ctr = int(seed) & 0x7FFFFFFF

    def _engine_forward(self, img, ids, seg, mask):
        if not img.is_cuda:
            raise L.MMVQAError("mm-vqa_amd runs on the GPU only: move the model and inputs to 'cuda' "
                               "(there is no CPU fallback by design)")
        d = self._desc
        img = img.contiguous().float()
        ids, seg, mask = (t.contiguous().long() for t in (ids, seg, mask))
        B, T = ids.shape
        self._ensure_plan(B, T, img.shape[2], img.shape[3])
        rows = B * T if d.head_kind == 0 else B
        V = d.n_classes
        ld = (V + 3) & ~3
        # (pad columns V..ld-1 are never read: the loss kernels mask them and the engine's GEMMs contract over V)
        buf = torch.empty(rows, ld, dtype=torch.float32, device=img.device)
        if ld != V:
            buf[:, V:].zero_()
        feat = torch.empty(B, d.feat_dim, dtype=torch.float32, device=img.device) if d.supcon else None
        self._seed_ctr = (self._seed_ctr * 1103515245 + 12345) & 0x7FFFFFFF
        L.check(L.lib().mmvqa_engine_forward(self._handle, L.stream_ptr(), L.ptr(img), L.ptr(ids), L.ptr(seg),
                                             L.ptr(mask), L.ptr(buf), ld, L.ptr(feat), 1 if self.training else 0,
                                             self._seed_ctr))
        self._fwd_state = (img, ids, seg, mask, rows, V, ld)
        logits = buf[:, :V]
        logits = logits.view(B, T, V) if d.head_kind == 0 else logits
        return (logits, feat) if d.supcon else logits

    def _engine_backward(self, dlogits, dfeat=None):
        img, ids, seg, mask, rows, V, ld = self._fwd_state
        first = next(p for n, p in self._params_by_name.items() if not n.startswith(_NEVER_USED))
        if first.grad is None:           # optimizer.zero_grad(set_to_none=True) semantics
            self._flat_grad.zero_()
        g = dlogits.reshape(rows, V) if dlogits.dim() == 3 else dlogits
        if not (g.stride(1) == 1 and g.stride(0) % 4 == 0 and g.stride(0) >= ld and g.data_ptr() % 16 == 0):
            pad = torch.zeros(rows, ld, dtype=torch.float32, device=g.device)
            pad[:, :V] = g
            g = pad
        gld = g.stride(0)
        if dfeat is not None:
            dfeat = dfeat.contiguous()
        L.check(L.lib().mmvqa_engine_backward(self._handle, L.stream_ptr(), L.ptr(g), gld, L.ptr(dfeat)))
        if getattr(self, "_cb_error", None) is not None:
            err, self._cb_error = self._cb_error, None
            raise err
        self.attach_grads()

    def forward(self, img, input_ids, segment_ids, input_mask):
        out = _ModelFn.apply(self, self._anchor, img, input_ids, segment_ids, input_mask)
        if self.dataset == "VQA-Med":
            return out, 0, 0   # models/mmbert.py:167
        return out

    # ------------------------------------------------------------------ data-parallel overlap
    def set_grad_ready_hook(self, fn, with_event=False):
        """fn(lo, hi) is called during backward as soon as flat_grads[lo:hi] is final (ordered on the current
        stream), in an order that partitions the whole buffer; pass None to remove.  Used by ddp.GradReducer to
        start the RCCL all-reduce of finished ranges while the backbone backward is still running.
        with_event=True: fn(lo, hi, ready) also receives a torch.cuda.Event recorded on the engine's stream at the
        announcement, i.e. after that stream has been ordered behind the side stream's weight gradients of the range
        (csrc/engine.cpp notify()): any other stream that waits on it may read the range."""
        self._cb_error = None
        if fn is None:
            self._cb = None
            L.check(L.lib().mmvqa_engine_set_grad_callback(self._handle, None, None))
            return

        def tramp(_user, lo, hi):
            try:
                if with_event:
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream())   # the stream backward was given: already behind the join
                    fn(int(lo), int(hi), ev)
                else:
                    fn(int(lo), int(hi))
            except BaseException as ex:   # exceptions cannot cross the C frame
                self._cb_error = ex

        self._cb = C.CFUNCTYPE(None, C.c_void_p, C.c_longlong, C.c_longlong)(tramp)
        L.check(L.lib().mmvqa_engine_set_grad_callback(self._handle, C.cast(self._cb, C.c_void_p), None))

    # ------------------------------------------------------------------ per-shape kernel tuning
    def tuned_persistent(self):
        """how many of the tuned GEMM shapes run in the persistent (stream-K) form"""
        return int(L.lib().mmvqa_engine_tune(self._handle, 2))

    def tune(self, img, input_ids, segment_ids, input_mask):
        """Time the candidate tile / split-K configurations of every GEMM shape of one training step
        (forward + backward on the given batch) and keep the fastest.  Parameters, BatchNorm buffers
        and gradients are left untouched (the pass itself computes garbage)."""
        lib = L.lib()
        bufs, nbt = self._flat[1].clone(), self._flat[2].clone()
        was_training = self.training
        self.train()
        L.check(min(0, lib.mmvqa_engine_tune(self._handle, 1)))
        cb = getattr(self, "_cb", None)
        if cb is not None:   # the throw-away pass must not trigger gradient all-reduces
            L.check(lib.mmvqa_engine_set_grad_callback(self._handle, None, None))
        try:
            out = self._engine_forward(img, input_ids, segment_ids, input_mask)
            logits = out[0] if isinstance(out, tuple) else out
            feat = out[1] if isinstance(out, tuple) else None
            self._engine_backward(torch.zeros_like(logits), None if feat is None else torch.zeros_like(feat))
            torch.cuda.synchronize()
        finally:
            n = lib.mmvqa_engine_tune(self._handle, 0)
            if cb is not None:
                L.check(lib.mmvqa_engine_set_grad_callback(self._handle, C.cast(cb, C.c_void_p), None))
            self._flat[1].copy_(bufs)
            self._flat[2].copy_(nbt)
            self._flat_grad.zero_()
            self.train(was_training)
        return n

    # ------------------------------------------------------------------ profiling (bench.py)
    def profile(self, enable, serialized: bool = False):
        """per-launch HIP-event timing of the next step; serialized=True puts every launch on one stream"""
        L.check(L.lib().mmvqa_engine_profile(self._handle, (2 if serialized else 1) if enable else 0))

    REGIONS = ("backbone", "tap", "qkv", "attention", "encoder_rest", "heads", "embed", "bn_coef", "qkv_attention_fused")
    # igemm_kernel | attention kernels | launches without matrix work | matrix work outside igemm_kernel (tapthin.hip, se.hip)
    PROFILE_CLASSES = ("igemm", "attention", "other", "matrix_other")

    def profile_read_regions(self):
        """{region: {class: {launches, ms, flops}}} of the profiled step (mmvqa_engine_profile_read_region)"""
        out = {}
        for r, rn in enumerate(self.REGIONS):
            out[rn] = {}
            for cls, nm in enumerate(self.PROFILE_CLASSES):
                n, ms, fl = C.c_longlong(), C.c_double(), C.c_double()
                L.check(L.lib().mmvqa_engine_profile_read_region(self._handle, r, cls, C.byref(n), C.byref(ms),
                                                                 C.byref(fl)))
                out[rn][nm] = dict(launches=n.value, ms=ms.value, flops=fl.value)
        return out

    HBM_KERNELS = ("bn_add_relu", "maxpool_fwd", "maxpool_bwd", "layernorm_fwd", "layernorm_bwd", "dropout_copy", "bn_act_add",
                   "dwconv_fwd", "dwconv_bwd_data", "dwconv_bwd_weight", "se_pool", "se_dgate", "act_bwd_stats",
                   "tap_thin_fwd", "tap_thin_bwd")

    def profile_read_hbm(self):
        """{kernel: {launches, ms, bytes}} of the HBM-bound kernels of the profiled step (mmvqa_engine_profile_read_hbm)"""
        out = {}
        for k, nm in enumerate(self.HBM_KERNELS, start=1):
            n, ms, by = C.c_longlong(), C.c_double(), C.c_double()
            L.check(L.lib().mmvqa_engine_profile_read_hbm(self._handle, k, C.byref(n), C.byref(ms), C.byref(by)))
            if n.value:
                out[nm] = dict(launches=n.value, ms=ms.value, bytes=by.value)
        return out

    def profile_read(self):
        out = {}
        for cls, nm in enumerate(self.PROFILE_CLASSES):
            n, ms, fl = C.c_longlong(), C.c_double(), C.c_double()
            L.check(L.lib().mmvqa_engine_profile_read(self._handle, cls, C.byref(n), C.byref(ms), C.byref(fl)))
            out[nm] = dict(launches=n.value, ms=ms.value, flops=fl.value)
        return out
