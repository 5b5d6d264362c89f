#!/bin/bash
cd /root/repo
export TBK_BENCH_SKIP_PEAK=1
for ov in 0 1; do
  TBK_H_OVERLAP_DIRECT=$ov python bench.py --cpu-sample 0 --config cfg2 --steps 4 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 overlap_direct=$ov', d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms_per_step'], d['max_abs_err_vs_oracle'])"
done
