#!/bin/bash
# Kernel traces of one-k calls at the BASELINE shapes:  bash tools/profile_single_k.sh r06 [cfg...]
# -> gpurun_out/<tag>_single_k/<cfg>/ ; tools/summarize_single_k.py turns them into profiles/<tag>_single_k_<cfg>.csv
TAG=${1:-r06}; shift
CFGS=${@:-cfg2 cfg3 cfg5}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in $CFGS; do
  O=$R/gpurun_out/${TAG}_single_k/$c
  mkdir -p $O
  python3 $R/tools/trace_single_k_cfg.py $c > $O/plain.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/tools/trace_single_k_cfg.py $c > $O/trace.log 2>&1
  cat $O/plain.log
  f=$(find $O/trace -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -12 "$f"
done
# FETCH_SIZE / WRITE_SIZE of the one-k kernels (one counter per pass, --kernel-trace only beside --pmc):  PMC=1 bash tools/profile_single_k.sh <tag> cfg2 cfg5
if [ "$PMC" = 1 ]; then
  for c in $CFGS; do
    O=$R/gpurun_out/${TAG}_single_k/$c
    for ctr in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$ctr -o t -- python3 $R/tools/trace_single_k_cfg.py $c 16 > $O/pmc_$ctr.log 2>&1
    done
    python3 - "$O" "$c" <<'PY'
import csv, glob, sys, collections
root, cfg = sys.argv[1:3]
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(root + "/pmc_%s/**/*counter_collection.csv" % ctr, recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == ctr:
                for key in ("hk_gemv_kernel", "hk_finish_wide_kernel", "hk_finish_kernel", "hk_csr_lds_kernel"):
                    if key in row["Kernel_Name"]:
                        acc[key].append(float(row["Counter_Value"]))
    for name, vals in sorted(acc.items()):
        print("%s %s %-24s launches %3d  mean %.5g KiB" % (cfg, ctr, name, len(vals), sum(vals) / len(vals)))
PY
  done
fi
