#!/bin/bash
cd /root/repo
export TBK_BENCH_SKIP_PEAK=1
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "cfg4 or mesh or fold or config4" 2>&1 | tail -4
for ov in 0 1; do
  TBK_H_OVERLAP=$ov python bench.py --cpu-sample 0 --config cfg4 --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 overlap=$ov', d['value'], d['ms_per_step'])"
done
