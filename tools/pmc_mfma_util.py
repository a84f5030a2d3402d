#!/usr/bin/env python3
"""MFMA utilisation of the GEMM launches of one training step from a rocprofv3 PMC pass.

    MMVQA_IGEMM_LOG=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES -d out --output-format csv -- \
        python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2> log.txt
    python tools/pmc_mfma_util.py out log.txt profiles/round2_mfma_util.json

utilisation = SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip's 1024 SIMDs) / (kernel duration x shader clock x 1024).
Under the counter pass launches run one at a time (no stream overlap), so this is the single-stream figure.  The launch
log (same run, same dispatch order) names the shape of every launch."""
import collections
import csv
import glob
import json
import re
import sys

CLK_GHZ, SIMDS = 2.4, 1024


def main():
    d, log, out = sys.argv[1:4]
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
    rows = [r for r in rows if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES"]
    ig = [r for r in rows if "igemm_kernel" in r["Kernel_Name"]]
    lines = [l for l in open(log) if l.startswith("igemm kind")]
    assert len(ig) == len(lines), (len(ig), len(lines))
    adam = [int(r["Dispatch_Id"]) for r in rows if "adam_kernel" in r["Kernel_Name"]]
    lo, hi = adam[-2], adam[-1]
    groups = collections.OrderedDict()

    def add(name, busy, ns, flops):
        g = groups.setdefault(name, [0, 0.0, 0.0, 0.0])
        g[0] += 1; g[1] += busy; g[2] += ns; g[3] += flops

    for r, l in zip(ig, lines):
        if not (lo < int(r["Dispatch_Id"]) <= hi):
            continue
        t = re.search(r"kind (\d) .* M (\d+) N (\d+) K (\d+) Cs (\d+) taps (\d+)", l)
        kind, M, N, K, Cs, taps = map(int, t.groups())
        busy = float(r["Counter_Value"])
        ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        fl = 2.0 * M * N * K
        add("all GEMM launches of the step", busy, ns, fl)
        qkv = (kind == 0 and M == 512 and N == 2304 and K == 768) or (kind == 1 and M == 512 and N == 768 and K == 2304) or \
              (kind == 2 and M == 2304 and N == 768 and K == 512)
        if qkv:
            add("fused QKV GEMM (fwd + dgrad + wgrad, 4 layers)", busy, ns, fl)
        if taps == 9 and Cs == 256:
            add("layer-3 3x3 convolutions (fwd + dgrad + wgrad)", busy, ns, fl)
        if M == 512 and N == 30522 or K == 30522 or M == 30522:
            add("vocabulary decoder GEMMs", busy, ns, fl)
    # the fused QKV projection + attention launches of the forward pass (qkvattn.hip) are not igemm_kernel launches
    for r in rows:
        if "qkv_attn_fwd_kernel" in r["Kernel_Name"] and lo < int(r["Dispatch_Id"]) <= hi:
            add("fused QKV projection + attention, forward (qkvattn.hip, 4 layers; B 16, T 32, 12 heads)", float(r["Counter_Value"]),
                int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), 2.0 * 512 * 2304 * 768 + 4.0 * 16 * 12 * 32 * 32 * 64)
    res = {}
    for k, (n, busy, ns, fl) in groups.items():
        res[k] = dict(launches=n, mfma_busy_cycles=busy, kernel_ms=ns / 1e6, mfma_util=busy / (ns * CLK_GHZ * SIMDS),
                      tflops=fl / ns / 1e3, frac_of_157_3=fl / ns / 1e3 / 157.3)
    res["note"] = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES, last training step of bench.py (config 2); util = busy cycles / "
                   "(duration x 2.4 GHz x 1024 SIMDs); launches run one at a time under the counter pass")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
