#!/usr/bin/env python3
"""Register / scratch / spill summary of every kernel in a gfx950 assembly file (hipcc -S --offload-device-only)."""
import re
import sys

txt = open(sys.argv[1]).read()
meta = txt[txt.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    blk = ".agpr_count:" + blk
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name").replace("_Z12igemm_kernel", "igemm").replace("Ev15mmvqa_gemm_desc7GemmAux", "")
    print(f"{name:60s} vgpr {g('vgpr_count'):>4s} agpr {g('agpr_count'):>4s} sgpr {g('sgpr_count'):>4s} "
          f"scratch {g('private_segment_fixed_size'):>5s} vspill {g('vgpr_spill_count'):>4s} sspill {g('sgpr_spill_count'):>4s}")
