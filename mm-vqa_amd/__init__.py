"""mm-vqa_amd: MI355X-native (gfx950) MMBERT training hot path of DannielSilva/MM-VQA.

Import as ``mmvqa_amd`` (the directory name carries a hyphen; the top-level ``mmvqa_amd`` package
points here).  All arithmetic is in libmmvqa_hip.so (mm-vqa_amd/csrc, C ABI in include/mmvqa.h).
"""
from . import _lib
from ._lib import MMVQAError
from .model import Model, desc_from_args
from .functional import mlm_loss, asl_loss, supcon_loss, split_feat
from .optim import FusedAdam
from . import synth

__all__ = ["Model", "desc_from_args", "mlm_loss", "asl_loss", "supcon_loss", "split_feat", "FusedAdam", "synth",
           "MMVQAError"]
