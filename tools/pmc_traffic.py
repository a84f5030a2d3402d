#!/usr/bin/env python3
"""Summarise the HBM traffic of the dominant kernel class from two rocprofv3 PMC passes
(--pmc FETCH_SIZE and --pmc WRITE_SIZE, collected separately: MI355X_MICROARCH.md 'rocprofv3 PMC slots').

gfx950 corrections (same guide, section HBM): FETCH_SIZE is reported in KiB and counts wide coalesced reads at
half their bytes (TCC_EA0_RDREQ x 64 B for 128-B requests) -> bytes = FETCH_SIZE * 1024 * 2;
WRITE_SIZE (KiB) is exact for 16-byte streaming stores and float atomics -> bytes = WRITE_SIZE * 1024.

    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/round1_igemm_traffic.json
"""
import csv
import glob
import json
import sys


def per_kernel(dirname, counter):
    """(kernel name, counter value) of every dispatch of the LAST training step (between the last two Adam launches)"""
    f = glob.glob(f"{dirname}/*/*_counter_collection.csv")[0]
    disp = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        disp.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], 0.0])[1] += float(r["Counter_Value"])
    ids = sorted(disp)
    adam = [i for i in ids if "adam_kernel" in disp[i][0]]
    lo, hi = adam[-2], adam[-1]
    return [disp[i] for i in ids if lo < i <= hi]


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    res = {}
    for cls, pat in (("igemm", "igemm_kernel"), ("all", "")):
        fk = [v for k, v in per_kernel(fetch_dir, "FETCH_SIZE") if pat in k]
        wk = [v for k, v in per_kernel(write_dir, "WRITE_SIZE") if pat in k]
        n = len(fk)
        fetch_b = sum(fk) * 1024.0 * 2.0
        write_b = sum(wk) * 1024.0
        res[cls] = dict(launches=n, fetch_bytes_per_launch=fetch_b / max(n, 1), write_bytes_per_launch=write_b / max(len(wk), 1),
                        hbm_bytes_per_launch=fetch_b / max(n, 1) + write_b / max(len(wk), 1),
                        hbm_bytes_total=fetch_b + write_b)
    res["note"] = ("last training step of `bench.py --steps 2 --warmup 1` under rocprofv3 --pmc (FETCH_SIZE and WRITE_SIZE "
                   "in separate passes); FETCH_SIZE x1024 x2 (gfx950 half-count correction), WRITE_SIZE x1024; "
                   "per launch = mean over the launches of the class in that step")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
