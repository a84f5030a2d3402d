#!/usr/bin/env python3
"""Where the chip is empty during a training step: a timeline digest of a `rocprofv3 --kernel-trace` run of bench.py.

The per-kernel statistics (tools/trace_steady.py) say how long each kernel runs; they do not say when nothing or only a
partial grid is resident.  This tool cuts the trace at the Adam launches (one per step), keeps the last N steps and reports
per step:
  * wall time with 0 / 1 / 2 / >=3 kernels resident (sweep over the start/end stamps of every dispatch);
  * wall time by the number of workgroups the live dispatches have between them (0, 1-31, 32-255, >= 256): with fewer than
    256 the chip cannot be full whatever the kernels do;
  * the largest gaps (intervals with NO kernel resident) with the kernels either side of them;
  * the time spent in launches whose grid cannot fill the chip, from the dispatch's own grid / workgroup / LDS / register
    figures: workgroups W, resident slots S = 256 CUs x (workgroups of that kernel one CU admits), rounds R = ceil(W / S),
    quantisation fill W / (R x S); `under` = duration x (1 - fill) is the share of that launch's CU-time without a
    workgroup if all its workgroups took equally long (a model, not a per-workgroup measurement);
  * the same split per kernel class.

    python tools/trace_timeline.py gpurun_out/prof/.../*_kernel_trace.csv profiles/round3_timeline_cfg2.json [N]
"""
import csv
import json
import math
import sys

CUS = 256
LDS_PER_CU = 160 * 1024
WAVES_PER_CU = 32
VGPR_BUDGET = 512      # per SIMD lane: unified VGPR+AGPR file, a wave with v registers leaves floor(512 / v) waves per SIMD


def slots_per_cu(r):
    threads = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    waves = max(1, (threads + 63) // 64)
    lds = int(r["LDS_Block_Size"])
    regs = int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"])
    by_waves = WAVES_PER_CU // waves
    by_lds = LDS_PER_CU // lds if lds > 0 else 99
    per_simd = max(1, VGPR_BUDGET // max(regs, 1))
    by_regs = (per_simd * 4) // waves
    return max(1, min(by_waves, by_lds, by_regs, 8))


def short(name):
    n = name.replace("void ", "")
    if n.startswith("igemm_kernel<"):
        a = n[len("igemm_kernel<"):n.index(">")].replace(" ", "").split(",")
        kind = {"0": "fwd", "1": "dgrad", "2": "wgrad"}.get(a[3], a[3])
        return f"igemm {kind} {a[0]}x{a[1]}x{a[2]}" + (" 8w" if a[5] == "2" else "") + (" nchw" if a[4] == "true" else "")
    return n.split("(")[0].split("<")[0]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    rows = sorted(csv.DictReader(open(src)), key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    lo, hi = adam[-nsteps - 1], adam[-1]
    win = rows[lo + 1:hi + 1]
    t0 = int(rows[lo]["End_Timestamp"])          # end of the previous step's Adam
    t1 = int(win[-1]["End_Timestamp"])
    wall = t1 - t0

    # ---- concurrency sweep
    ev = []
    for i, r in enumerate(win):
        ev.append((int(r["Start_Timestamp"]), 1, i))
        ev.append((int(r["End_Timestamp"]), -1, i))
    ev.sort(key=lambda e: (e[0], e[1]))
    conc = [0, 0, 0, 0]
    gaps = []
    live, prev_t, last_ended = 0, t0, None
    wgs = []
    for r in win:
        threads = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        wgs.append((int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])) // max(threads, 1))
    demand = 0                       # workgroups of the dispatches that are live (what the chip COULD be running)
    dem = {"0": 0, "1-31": 0, "32-255": 0, ">=256": 0}
    for t, d, i in ev:
        conc[min(live, 3)] += t - prev_t
        dem["0" if demand == 0 else "1-31" if demand < 32 else "32-255" if demand < 256 else ">=256"] += t - prev_t
        demand += d * wgs[i]
        if live == 0 and t > prev_t and d == 1:
            gaps.append((t - prev_t, last_ended, i))
        prev_t = t
        live += d
        if d == -1:
            last_ended = i
    gaps.sort(key=lambda g: -g[0])
    gap_total = sum(g[0] for g in gaps)
    hist = {}
    for g in gaps:
        b = "<1us" if g[0] < 1000 else "1-2us" if g[0] < 2000 else "2-4us" if g[0] < 4000 else "4-8us" if g[0] < 8000 else ">=8us"
        h = hist.setdefault(b, [0, 0])
        h[0] += 1; h[1] += g[0]

    # ---- which pairs of kernels the gaps sit between
    pair = {}
    for ns, a, b in gaps:
        k = (short(win[a]["Kernel_Name"]) if a is not None else "(step start)") + " -> " + short(win[b]["Kernel_Name"])
        p = pair.setdefault(k, [0, 0])
        p[0] += 1; p[1] += ns

    # ---- grids that cannot fill the chip
    cls = {}
    under_total = 0.0
    small = []
    for r in win:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        threads = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        W = (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])) // max(threads, 1)
        spc = slots_per_cu(r)
        S = CUS * spc
        # a grid below one workgroup per CU leaves CUs empty; above it, the last round is partial
        if W <= CUS:
            fill = W / CUS
        else:
            R = math.ceil(W / S)
            fill = W / (R * S) if R > 1 else 1.0   # one round with >= 256 workgroups: every CU has work
        under = d * (1.0 - fill)
        under_total += under
        c = cls.setdefault(short(r["Kernel_Name"]), dict(launches=0, ns=0, under_ns=0.0, lt256=0, lt256_ns=0))
        c["launches"] += 1; c["ns"] += d; c["under_ns"] += under
        if W < CUS:
            c["lt256"] += 1; c["lt256_ns"] += d
            small.append((d, short(r["Kernel_Name"]), W))
    small.sort(key=lambda s: -s[0])
    ksum = sum(c["ns"] for c in cls.values())

    out = dict(
        source=src, steps=nsteps, launches_per_step=len(win) / nsteps,
        wall_ms_per_step=wall / 1e6 / nsteps,
        kernel_ms_per_step=ksum / 1e6 / nsteps,
        resident_kernels_ms_per_step={"0": conc[0] / 1e6 / nsteps, "1": conc[1] / 1e6 / nsteps, "2": conc[2] / 1e6 / nsteps,
                                      ">=3": conc[3] / 1e6 / nsteps},
        average_concurrency=ksum / max(1, wall - conc[0]),
        live_workgroup_demand_ms_per_step={k: v / 1e6 / nsteps for k, v in dem.items()},
        gaps=dict(count_per_step=len(gaps) / nsteps, total_ms_per_step=gap_total / 1e6 / nsteps,
                  histogram={k: dict(count_per_step=v[0] / nsteps, ms_per_step=v[1] / 1e6 / nsteps) for k, v in hist.items()},
                  by_neighbours=[dict(between=k, count_per_step=v[0] / nsteps, ms_per_step=v[1] / 1e6 / nsteps)
                                 for k, v in sorted(pair.items(), key=lambda kv: -kv[1][1])[:20]],
                  largest=[dict(us=ns / 1e3, after=short(win[a]["Kernel_Name"]) if a is not None else "(step start)",
                                before=short(win[b]["Kernel_Name"])) for ns, a, b in gaps[:20]]),
        partial_grids=dict(
            note="under_ms = sum over launches of duration x (1 - W / (rounds x resident slots)), W < 256 counted against 256 CUs",
            under_ms_per_step=under_total / 1e6 / nsteps,
            launches_below_256_workgroups_per_step=sum(c["lt256"] for c in cls.values()) / nsteps,
            ms_in_launches_below_256_workgroups_per_step=sum(c["lt256_ns"] for c in cls.values()) / 1e6 / nsteps,
            largest_below_256=[dict(us=d / 1e3, kernel=k, workgroups=W) for d, k, W in small[:20]],
            by_class=[dict(kernel=k, launches_per_step=c["launches"] / nsteps, ms_per_step=c["ns"] / 1e6 / nsteps,
                           under_ms_per_step=c["under_ns"] / 1e6 / nsteps,
                           below_256_launches_per_step=c["lt256"] / nsteps, below_256_ms_per_step=c["lt256_ns"] / 1e6 / nsteps)
                      for k, c in sorted(cls.items(), key=lambda kv: -kv[1]["ns"])[:30]]))
    json.dump(out, open(dst, "w"), indent=1)
    r = out["resident_kernels_ms_per_step"]
    print(f"{nsteps} steps: wall {out['wall_ms_per_step']:.2f} ms/step; resident 0/1/2/3+: {r['0']:.2f} / {r['1']:.2f} / "
          f"{r['2']:.2f} / {r['>=3']:.2f} ms; {out['gaps']['count_per_step']:.0f} gaps = {out['gaps']['total_ms_per_step']:.2f} ms; "
          f"partial-grid under-fill {out['partial_grids']['under_ms_per_step']:.2f} ms; live workgroups 0 / 1-31 / 32-255 / 256+: "
          + " / ".join(f"{v:.2f}" for v in out["live_workgroup_demand_ms_per_step"].values()) + " ms")


if __name__ == "__main__":
    main()
