#!/bin/bash
# Kernel trace of the eigenvalue path at the orbital counts between the BASELINE configs (register cascade of DESIGN §5.5):
#   bash tools/profile_sizes.sh r03      (through gpurun, from the repository root)
# -> gpurun_out/<tag>/trace_sizes/ ; copy the *_kernel_stats.csv into profiles/<tag>_kernel_stats_sizes.csv
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_sizes -o sizes -- python3 $R/tools/bench_sizes.py 80 96 128 160 > $O/trace_sizes.log 2>&1
tail -4 $O/trace_sizes.log
find $O/trace_sizes -name "*kernel_stats*" | head -2
