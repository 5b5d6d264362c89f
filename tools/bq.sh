#!/bin/bash
cd /root/repo
export TBK_BENCH_SKIP_PEAK=1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
for c in cfg4 cfg2; do
  python bench.py --cpu-sample 0 --config $c --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', d['value'], d['ms_per_step'])"
done
mkdir -p gpurun_out/tl4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tl4 -- python3 bench.py --cpu-sample 0 --config cfg4 --steps 1 --warmup 1 > gpurun_out/tl4.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/tl4/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:4]: print(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e6)
PY
