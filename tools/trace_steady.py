#!/usr/bin/env python3
"""Per-kernel statistics of the LAST training steps of a `rocprofv3 --kernel-trace` run of bench.py.

The whole-run `*_kernel_stats.csv` of rocprofv3 also contains the launcher's one-off tuning pass (every candidate tile
of every GEMM shape, some of them deliberately bad), so its per-kernel averages are not those of a training step.
This tool cuts the trace at the Adam launches (one per step), keeps the last N steps and writes the same columns.

    python tools/trace_steady.py gpurun_out/prof/.../*_kernel_trace.csv profiles/round1_c_kernel_stats_steady.csv [N]
"""
import collections
import csv
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    rows = sorted(csv.DictReader(open(src)), key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    lo, hi = adam[-nsteps - 1], adam[-1]
    win = rows[lo + 1:hi + 1]
    agg = collections.OrderedDict()
    for r in win:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg.setdefault(r["Kernel_Name"], [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    wall = int(win[-1]["End_Timestamp"]) - int(win[0]["Start_Timestamp"])
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, a[0], a[1], f"{a[1] / a[0]:.1f}", f"{100.0 * a[1] / tot:.2f}", a[2], a[3]])
        w.writerow([f"# {nsteps} steps: wall {wall / 1e6:.3f} ms, sum of kernel durations {tot / 1e6:.3f} ms "
                    f"(two HIP streams overlap), {len(win)} launches", "", "", "", "", "", ""])
    ig = [(k, a) for k, a in agg.items() if "igemm_kernel" in k]
    n = sum(a[0] for _, a in ig); t = sum(a[1] for _, a in ig)
    print(f"{nsteps} steps: wall {wall / 1e6 / nsteps:.2f} ms/step, kernels {tot / 1e6 / nsteps:.2f} ms/step, "
          f"igemm {n // nsteps} launches/step avg {t / n / 1e3:.1f} us")


if __name__ == "__main__":
    main()
