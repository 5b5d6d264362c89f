#!/bin/bash
cd /root/repo
export TBK_BENCH_SKIP_PEAK=1
for pr in 0 1; do
for c in cfg4 cfg2; do
  TBK_MAIN_PRIO=$pr python bench.py --cpu-sample 0 --config $c --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c prio=$pr', d['value'], d['ms_per_step'])"
done
done
