#!/usr/bin/env python3
"""
Turn the rocprofv3 outputs of one round into the small, tracked files under profiles/.

    python tools/summarize_profiles.py <round tag> <gpurun_out dir with *_kernel_stats.csv / *_counter_collection.csv>

Expects (any subset):  <dir>/trace/*_kernel_stats.csv      rocprofv3 --kernel-trace --stats -- python3 bench.py
                       <dir>/fetch/*_counter_collection.csv rocprofv3 --pmc FETCH_SIZE --kernel-trace -- ...
                       <dir>/write/*_counter_collection.csv rocprofv3 --pmc WRITE_SIZE --kernel-trace -- ...
                       <dir>/sq/*_counter_collection.csv    rocprofv3 --pmc SQ_* --kernel-trace -- ...
                       <dir>/sq2/*_counter_collection.csv   (a second SQ group: the counters do not fit one pass)
Writes profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc_summary.txt and profiles/<tag>_traffic.json
(HBM bytes per launch of the H(k) kernel, the number bench.py reports as roofline.traffic).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

KERNELS = ("hk_dense", "hk_csr", "herm_tridiag", "phase_rows", "tridiag_ql")


def load_counters(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for name in glob.glob(os.path.join(path, "*_counter_collection.csv")):
        with open(name) as handle:
            for row in csv.DictReader(handle):
                for key in KERNELS:
                    if key in row["Kernel_Name"]:
                        agg[key][row["Counter_Name"]].append((float(row["Counter_Value"]), int(row["Grid_Size"])))
    return agg


def full_launch_average(vals):
    """Average over the launches with the largest grid (= full k chunks)."""
    top = max(g for _, g in vals)
    big = [v for v, g in vals if g == top]
    return sum(big) / len(big), len(big), top


def main():
    tag, src = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    for name in glob.glob(os.path.join(src, "trace", "*_kernel_stats.csv")):
        shutil.copy(name, os.path.join(out_dir, "%s_kernel_stats.csv" % tag))
    lines = [
        "%s -- PMC summary.  rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python3 bench.py "
        "--cpu-sample 0 --steps 1 --warmup 0 (one pass per counter group)" % tag,
        "values: average per launch over the full-size k chunks (32768 k-points of cfg2: N_orb=64, N_R=4096)",
        "FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced",
        "stream (MI355X_MICROARCH.md, HBM), so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE matched a known byte",
        "count here (phase rows: 2.147 GB expected, 2.147 GB counted) and is used as is.",
        "",
    ]
    traffic = {}
    for group in ("fetch", "write", "sq", "sq2"):
        agg = load_counters(os.path.join(src, group))
        for kernel in KERNELS:
            for counter, vals in sorted(agg.get(kernel, {}).items()):
                avg, count, grid = full_launch_average(vals)
                lines.append("%-14s %-30s %16.6g   (launches averaged: %d, grid %d)" % (kernel, counter, avg, count, grid))
                if counter in ("FETCH_SIZE", "WRITE_SIZE"):
                    traffic.setdefault(kernel, {})[counter] = avg
    for kernel, vals in traffic.items():
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            vals["hbm_bytes_per_launch"] = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
            vals["kpoints_per_launch"] = 32768
            lines.append("%-14s HBM bytes per launch = (2 x FETCH + WRITE) x 1024 = %.4g" % (kernel, vals["hbm_bytes_per_launch"]))
    with open(os.path.join(out_dir, "%s_pmc_summary.txt" % tag), "w") as handle:
        handle.write("\n".join(lines) + "\n")
    with open(os.path.join(out_dir, "%s_traffic.json" % tag), "w") as handle:
        json.dump(traffic, handle, indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
