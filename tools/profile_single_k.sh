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
