#!/bin/bash
cd /root/repo
export TBK_BENCH_SKIP_PEAK=1
for kc in 0 31744 29696 27776 25728 33792; do
  python bench.py --cpu-sample 0 --config cfg2 --steps 4 --warmup 1 --k-chunk $kc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 k_chunk=$kc', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['launches'])"
done
