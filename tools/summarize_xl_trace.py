#!/usr/bin/env python3
"""Summary of the rocprofv3 runs of tools/bench_xl.py (own path above 1024 orbitals) for profiles/:

    python tools/summarize_xl_trace.py <kernel-trace dir> [<FETCH_SIZE dir> <WRITE_SIZE dir>]  > profiles/rNN_xl_<n>.txt

The trace dir holds t_kernel_trace.csv of `rocprofv3 --kernel-trace --stats -- python3 tools/bench_xl.py --own <n>`; the counter dirs
t_counter_collection.csv of `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 tools/bench_xl.py --own --nk64 <n>`.
Per kernel and batch size (the grid tells the batch): launches, total and average duration; per kernel: the counters' totals in bytes
(KiB counters; FETCH_SIZE doubled for gfx950 as MI355X_MICROARCH.md prescribes)."""
import collections
import csv
import os
import sys

NAMES = ("band_xl_sweep_kernel", "band_xl_sweep4_kernel", "band_xl_xsum_kernel", "band_xl_serial_kernel", "band_xl_update_kernel",
         "band_chase4g_kernel", "band_chase4w_kernel", "tridiag_bisect_kernel", "band_extract_from_kernel", "band_deposit_kernel")


def short(name):
    for n in NAMES:
        if n in name:
            return n
    return None


def main():
    trace = list(csv.DictReader(open(os.path.join(sys.argv[1], "t_kernel_trace.csv"))))
    agg = collections.defaultdict(lambda: [0, 0])
    for r in trace:
        k = short(r["Kernel_Name"])
        if k is None:
            continue
        wg = int(r["Workgroup_Size_X"])
        gy = int(r["Grid_Size_Y"])
        # matrices of the launch: blockIdx.y for the 2-D launches, blockIdx.x (grid / workgroup size) for the others
        mats = gy if ("sweep" in k or "xsum" in k or "update" in k) else int(r["Grid_Size_X"]) // wg
        a = agg[(k, mats)]
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print("%-28s %9s %9s %12s %12s" % ("kernel", "matrices", "launches", "total ms", "average us"))
    for (k, mats), (cnt, ns) in sorted(agg.items()):
        print("%-28s %9d %9d %12.3f %12.2f" % (k, mats, cnt, ns / 1e6, ns / cnt / 1e3))
    if len(sys.argv) >= 4:
        print()
        for label, d, factor in (("FETCH_SIZE (x 2 x 1024 B)", sys.argv[2], 2048.0), ("WRITE_SIZE (x 1024 B)", sys.argv[3], 1024.0)):
            tot = collections.defaultdict(float)
            cnt = collections.Counter()
            for r in csv.DictReader(open(os.path.join(d, "t_counter_collection.csv"))):
                k = short(r["Kernel_Name"])
                if k:
                    tot[k] += float(r["Counter_Value"]) * factor
                    cnt[k] += 1
            for k in sorted(tot):
                print("%-28s %-26s %6d launches %10.2f GB  (%.1f MB per launch)" % (k, label, cnt[k], tot[k] / 1e9, tot[k] / cnt[k] / 1e6))


if __name__ == "__main__":
    main()
