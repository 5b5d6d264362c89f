#!/usr/bin/env python3
"""
Turn the rocprofv3 outputs of one round (tools/profile_round.sh) into the small, tracked files under profiles/.

    python tools/summarize_profiles.py <round tag> <gpurun_out dir>

Expects (any subset):  <dir>/trace/*_kernel_stats.csv, <dir>/trace_cfgN/*_kernel_stats.csv     kernel time summaries
                       <dir>/cfgN_{fetch,write,sq,sq2,tcc}/*_counter_collection.csv            one PMC group per pass
Writes profiles/<tag>_kernel_stats[_cfgN].csv, profiles/<tag>_pmc_summary.txt (cfg2, the bench line),
profiles/<tag>_pmc_cfg3.txt, _cfg4.txt, _cfg5.txt and profiles/<tag>_traffic.json (HBM bytes per launch of the
H(k) kernel: what bench.py reports as roofline.traffic).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

KERNELS = ("hk_dense", "hk_csr", "herm_tridiag4", "herm_tridiag8", "herm_tridiag_packed", "herm_tridiag_stream", "band_reduce",
           "band_chase", "band_extract", "phase_rows", "tridiag_ql", "tridiag_bisect", "fold_rows", "gather_place")
CONFIG_NOTE = {
    "cfg2": "cfg2: dense N_orb=64, N_R=4096, 100 000 random k-points (the bench line)",
    "cfg3": "cfg3: CSR N_orb=256, N_R=512, 50 000 random k-points",
    "cfg4": "cfg4: the cfg2 model on the 100 x 100 x 100 mesh (folded evaluation), one GPU",
    "cfg5": "cfg5: dense N_orb=512, N_R=2048, 10 000 random k-points",
}


def load_counters(path):
    """kernel -> counter -> [(dispatch id, value, grid)] for every launch of the profiled command."""
    series = collections.defaultdict(lambda: collections.defaultdict(list))
    for name in glob.glob(os.path.join(path, "*_counter_collection.csv")):
        with open(name) as handle:
            for row in csv.DictReader(handle):
                for key in KERNELS:
                    if key in row["Kernel_Name"]:
                        series[key][row["Counter_Name"]].append(
                            (int(row.get("Dispatch_Id", 0) or 0), float(row["Counter_Value"]), int(row["Grid_Size"])))
                        break
    return series


def full_launch_average(vals):
    """Average over the launches with the largest grid (= full k chunks) of the MEASURED step: the profiled command runs one
    warm-up step and one measured step with the same full-size launches, so of those -- in dispatch order -- the first
    half belongs to the warm-up and is dropped.  (The one-k calls of the line's host_api section are many small launches
    behind the measured step: they never have the largest grid.)"""
    top = max(g for _, _, g in vals)
    big = sorted((d, v) for d, v, g in vals if g == top)
    measured = big[len(big) // 2:] if len(big) >= 2 else big
    return sum(v for _, v in measured) / len(measured), len(measured), top


def summarize_config(tag, src, cfg):
    lines = [
        "%s -- PMC summary, %s." % (tag, CONFIG_NOTE[cfg]),
        "rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python3 bench.py --cpu-sample 0 --config %s "
        "--steps 1 --warmup 1 (one pass per counter group; the launches of the warm-up step are dropped: the measured step only)" % cfg,
        "values: average per launch over the launches with the largest grid (= full k chunks)",
        "FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced",
        "stream (MI355X_MICROARCH.md, HBM), so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE matched a known byte",
        "count (phase rows, round 1) and is used as is.  SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles.",
        "",
    ]
    traffic = {}
    found = False
    for group in ("fetch", "write", "sq", "sq2", "tcc"):
        agg = load_counters(os.path.join(src, "%s_%s" % (cfg, group)))
        for kernel in KERNELS:
            for counter, vals in sorted(agg.get(kernel, {}).items()):
                found = True
                avg, count, grid = full_launch_average(vals)
                lines.append("%-20s %-30s %16.6g   (launches averaged: %d, grid %d)" % (kernel, counter, avg, count, grid))
                if counter in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"):
                    traffic.setdefault(kernel, {})[counter] = avg
    for kernel, vals in traffic.items():
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            vals["hbm_bytes_per_launch"] = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
            lines.append("%-20s HBM bytes per launch = (2 x FETCH + WRITE) x 1024 = %.4g" % (kernel, vals["hbm_bytes_per_launch"]))
        if "TCC_HIT_sum" in vals and "TCC_MISS_sum" in vals:
            vals["l2_hit_rate"] = vals["TCC_HIT_sum"] / max(1.0, vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"])
            lines.append("%-20s L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS) = %.3f" % (kernel, vals["l2_hit_rate"]))
    return (lines if found else None), traffic


def main():
    tag, src = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    for name in glob.glob(os.path.join(src, "trace", "*_kernel_stats.csv")):
        shutil.copy(name, os.path.join(out_dir, "%s_kernel_stats.csv" % tag))
    for cfg in ("cfg1", "cfg3", "cfg4", "cfg5"):
        for name in glob.glob(os.path.join(src, "trace_%s" % cfg, "*_kernel_stats.csv")):
            shutil.copy(name, os.path.join(out_dir, "%s_kernel_stats_%s.csv" % (tag, cfg)))
    for cfg in ("cfg2", "cfg3", "cfg4", "cfg5"):
        lines, traffic = summarize_config(tag, src, cfg)
        if lines is None:
            continue
        name = "%s_pmc_summary.txt" % tag if cfg == "cfg2" else "%s_pmc_%s.txt" % (tag, cfg)
        with open(os.path.join(out_dir, name), "w") as handle:
            handle.write("\n".join(lines) + "\n")
        if cfg == "cfg2":
            if "hk_dense" in traffic:
                traffic["hk_dense"]["kpoints_per_launch"] = 32768  # the averaged launches are the full 32768-k chunks
            with open(os.path.join(out_dir, "%s_traffic.json" % tag), "w") as handle:
                json.dump(traffic, handle, indent=1, sort_keys=True)
        print("\n".join(lines))


if __name__ == "__main__":
    main()
