"""CPU checks of the drop-in boundary: the shared library loads, exports every symbol that
include/mmvqa.h declares, and the ctypes mirror agrees with the header (argument counts, struct sizes).
No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "mmvqa.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = src[src.index("typedef struct mmvqa_engine mmvqa_engine;"):]
    out = {}
    for m in re.finditer(r"\b([A-Za-z_][\w\s\*]*?)\b(mmvqa_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        args = m.group(3).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        out[m.group(2)] = n
    return out


def test_library_exports_every_declared_symbol():
    from mmvqa_amd import _lib as L
    lib = L.lib()   # raises if a declared symbol is missing (getattr on the CDLL)
    decl = header_functions()
    assert len(decl) >= 40
    for name, nargs in decl.items():
        assert hasattr(lib, name), f"{name} declared in include/mmvqa.h but not exported"
        assert name in L.SIGNATURES, f"{name} has no ctypes signature"
        assert len(L.SIGNATURES[name][1]) == nargs, f"{name}: header has {nargs} args, ctypes {len(L.SIGNATURES[name][1])}"
    assert set(L.SIGNATURES) == set(decl)
    assert lib.mmvqa_version() >= 100
    assert lib.mmvqa_sizeof_gemm_desc() == C.sizeof(L.GemmDesc)
    assert lib.mmvqa_sizeof_attn_desc() == C.sizeof(L.AttnDesc)
    assert lib.mmvqa_sizeof_model_desc() == C.sizeof(L.ModelDesc)


def test_error_reporting_without_gpu():
    from mmvqa_amd import _lib as L
    lib = L.lib()
    d = L.ModelDesc()   # all zeros: invalid
    h = C.c_void_p()
    rc = lib.mmvqa_engine_create(C.byref(d), C.byref(h))
    assert rc != 0 and b"engine_create" in lib.mmvqa_last_error()
    with pytest.raises(L.MMVQAError):
        L.check(rc)


def test_igemm_refuses_inconsistent_descriptors():
    """misuse returns MMVQA_ERR_ARG on the host instead of reading outside the caller's buffers on the device
    (the loaders address operands from the descriptor's dimensions alone)"""
    from mmvqa_amd import _lib as L
    lib = L.lib()

    def desc():
        d = L.GemmDesc()
        d.M, d.N, d.K = 64, 64, 64
        d.A, d.B, d.C = 0x1000, 0x2000, 0x3000          # never dereferenced: every case below is refused first
        d.a_ld, d.b_ld, d.c_ld, d.g_Cs = 64, 64, 64, 64
        d.g_SH = d.g_SW = d.g_OH = d.g_OW = 1
        d.g_KH = d.g_KW = 1
        d.g_stride, d.g_pad = 1, 0
        return d

    def refused(d, kind, what):
        rc = lib.mmvqa_igemm(C.byref(d), kind, 0, 0, None)
        assert rc == -1, f"{what}: rc {rc}"
        assert b"igemm:" in lib.mmvqa_last_error(), what

    d = desc(); d.K = 96                       # K != taps * Cs
    refused(d, L.KIND_FWD, "K mismatch")
    d = desc(); d.a_ld = 32                    # rows shorter than the contraction
    refused(d, L.KIND_FWD, "a_ld")
    d = desc(); d.b_ld = 8
    refused(d, L.KIND_FWD, "b_ld")
    d = desc(); d.A = None
    refused(d, L.KIND_FWD, "null A")
    d = desc(); d.a_pro = L.PRO_DZ             # BatchNorm-backward prologue without its second tensor / coefficients
    refused(d, L.KIND_DGRAD, "PRO_DZ without A2")
    d = desc(); d.g_KH = d.g_KW = 3; d.g_pad = 1; d.g_SH = d.g_SW = d.g_OH = d.g_OW = 8; d.M = 100   # M not whole images
    d.K = 9 * 64; d.b_ld = 9 * 64
    refused(d, L.KIND_FWD, "M not a multiple of OH*OW")
    d = desc(); d.N = 100                      # wgrad: N != taps * Cs
    refused(d, L.KIND_WGRAD, "wgrad N")
    d = desc(); d.dact = L.ACT_GELU            # act' epilogue without the saved pre-activation
    refused(d, L.KIND_DGRAD, "dact without Pre")
    d = desc(); d.drop_p = 1.5
    refused(d, L.KIND_FWD, "dropout p")
    refused(desc(), 7, "unknown kind")


def test_missing_library_fails_loudly(monkeypatch):
    from mmvqa_amd import _lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libmmvqa_hip.so")
    with pytest.raises(L.MMVQAError, match="no CPU fallback"):
        L.lib()


def test_model_protocol_on_cpu():
    """Model(args) builds its parameter table from the engine without a GPU: state_dict names/shapes equal
    the reference's (via the oracle classes, which are pinned to the reference), ResNet-152 has the
    published parameter count, unknown options raise NotImplementedError, forward on CPU refuses."""
    import torch
    import mmvqa_amd
    from oracle import mmbert_oracle as O
    args = O.make_args(resnet_layers=(1, 1, 1, 1), resnet_width=8, hidden_size=96, n_layers=2, heads=12,
                       vocab_size=50, emb_vocab=50, bert_max_pos=32)
    for tm, ds, sc in (("transformer", "roco", False), ("realformer", "roco", True), ("transformer", "VQA-Med", False)):
        a = O.make_args(**{**vars(args), "transformer_model": tm, "dataset": ds, "supcon": sc})
        m = mmvqa_amd.Model(a)
        o = O.OracleModel(a)
        sd, osd = m.state_dict(), o.state_dict()
        assert set(sd) == set(osd)
        assert all(sd[k].shape == osd[k].shape for k in sd)
        m.load_state_dict(osd)
        assert all(torch.equal(m.state_dict()[k], osd[k]) for k in osd)
    full = mmvqa_amd.Model(O.make_args())
    n_backbone = sum(p.numel() for n, p in full.named_parameters() if n.startswith("transformer.trans.model."))
    assert n_backbone == 60192808                       # torchvision resnet152
    assert sum(p.numel() for p in full.parameters()) == 140_024_674   # = the oracle's (reference-pinned) Model
    # timm tf_efficientnetv2_m(features_only=True): published body size, oracle-identical state_dict
    eff = mmvqa_amd.Model(O.make_args(cnn_encoder="tf_efficientnetv2_m", transformer_model="realformer"))
    n_eff = sum(p.numel() for n, p in eff.named_parameters() if n.startswith("transformer.trans.model."))
    assert n_eff == 52200436
    ea = O.make_args(cnn_encoder="tf_efficientnetv2_m", effnet_depth_div=8, hidden_size=96, n_layers=1, vocab_size=50,
                     emb_vocab=50, bert_max_pos=32)
    esd, eosd = mmvqa_amd.Model(ea).state_dict(), O.OracleModel(ea).state_dict()
    assert set(esd) == set(eosd) and all(esd[k].shape == eosd[k].shape for k in esd)
    with pytest.raises(NotImplementedError):
        mmvqa_amd.Model(O.make_args(transformer_model="lstm"))
    with pytest.raises(NotImplementedError):
        mmvqa_amd.Model(O.make_args(cnn_encoder="vgg16"))
    with pytest.raises(mmvqa_amd.MMVQAError):
        m(torch.zeros(1, 3, 32, 32), torch.zeros(1, 8, dtype=torch.long), torch.zeros(1, 8, dtype=torch.long),
          torch.ones(1, 8, dtype=torch.long))


def test_comm_library_exports_every_declared_symbol():
    """include/mmvqa_comm.h (the RCCL wrappers of SURVEY 8(b)): libmmvqa_comm.so loads and exports every entry point;
    no collective is issued (no GPU here)"""
    src = open(os.path.join(ROOT, "include", "mmvqa_comm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(mmvqa_\w+)\s*\(", src))
    assert {"mmvqa_allreduce_bucket", "mmvqa_allgather", "mmvqa_broadcast", "mmvqa_comm_create", "mmvqa_comm_unique_id"} <= names
    from mmvqa_amd.ddp import NativeComm
    lib = NativeComm.lib()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mmvqa_comm.h but not exported"
    assert lib.mmvqa_comm_rccl_version() > 20000
