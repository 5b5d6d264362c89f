#!/bin/bash
# quick A/B on the GPU box: the bench line of one config with an environment switch off and on
#   bash tools/bq.sh cfg4 TBK_H_OVERLAP        (through gpurun, from the repository root)
cd "${GRAFT_REPO_ROOT:-.}"
export TBK_BENCH_SKIP_PEAK=1
CFG=${1:-cfg2}; VAR=${2:-TBK_H_OVERLAP}
for v in 0 1; do
  env $VAR=$v python bench.py --cpu-sample 0 --config $CFG --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$CFG $VAR=$v', d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
done
