#!/usr/bin/env python3
"""Ranks the GEMM shapes of one training step by kernel time: joins the launcher's per-launch log (MMVQA_IGEMM_LOG=1,
stderr) with a rocprofv3 --kernel-trace CSV of the SAME run by dispatch order.

    MMVQA_IGEMM_LOG=1 rocprofv3 --kernel-trace -d out -o t --output-format csv -- python3 bench.py --steps 2 --warmup 1 \
        --no-cpu-baseline --no-roofline 2> log.txt
    python tools/igemm_rank.py out/t_kernel_trace.csv log.txt [top]"""
import collections
import csv
import re
import sys


def main():
    trace, log = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Dispatch_Id"]))
    ig = [r for r in rows if "igemm_kernel" in r["Kernel_Name"]]
    lines = [l for l in open(log) if l.startswith("igemm kind")]
    assert len(ig) == len(lines), (len(ig), len(lines))
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    lo, hi = int(rows[adam[-2]]["Dispatch_Id"]), int(rows[adam[-1]]["Dispatch_Id"])
    agg = collections.OrderedDict()
    other = collections.Counter()
    for r in rows:
        if lo < int(r["Dispatch_Id"]) <= hi and "igemm_kernel" not in r["Kernel_Name"]:
            other[r["Kernel_Name"].split("(")[0][-48:]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for r, l in zip(ig, lines):
        if not (lo < int(r["Dispatch_Id"]) <= hi):
            continue
        t = re.search(r"kind (\d) fast (\d) tile (\S+) ks (\d) M (\d+) N (\d+) K (\d+) Cs (\d+) taps (\d+) stride (\d) apro (\d) bpro (\d) splitk (\d+)", l)
        kind, fast, tile, ks, M, N, K, Cs, taps, stride, apro, bpro, sk = t.groups()
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        key = (int(kind), int(fast), int(M), int(N), int(K), int(Cs), int(taps), int(stride), int(apro), int(bpro), tile, int(ks), int(sk))
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += d
    tot = sum(a[1] for a in agg.values())
    fl = sum(2.0 * k[2] * k[3] * k[4] * a[0] for k, a in agg.items())
    print(f"last step: igemm {sum(a[0] for a in agg.values())} launches, {tot / 1e3:.2f} ms, {fl / tot / 1e6:.1f} TFLOP/s; "
          f"other kernels {sum(other.values()) / 1e3:.2f} ms")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        kind, fast, M, N, K, Cs, taps, stride, apro, bpro, tile, ks, sk = k
        print(f"{a[1]:8.0f} us x{a[0]:3d} {a[1] / a[0]:7.1f} us {2.0 * M * N * K * a[0] / a[1] / 1e6:6.1f} TF  kind {kind} fast {fast} "
              f"M {M} N {N} K {K} Cs {Cs} taps {taps} s{stride} apro {apro} bpro {bpro} tile {tile} ks {ks} splitk {sk}")
    print("other kernels:")
    for k, v in other.most_common(12):
        print(f"{v:8.0f} us  {k}")


if __name__ == "__main__":
    main()
