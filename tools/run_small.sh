# measurement helper (GPU box): time of the n <= 64 reduction kernel per batch, e.g.  bash tools/run_small.sh "64 32768" "48 65536"
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "$@"; do
  set -- $cfg
  rm -rf /tmp/tr; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o t -- python3 $R/tools/race_check.py $1 $2 4 > /tmp/tr.log 2>&1
  tail -1 /tmp/tr.log
  python3 - "$1" "$2" <<'PY'
import csv, sys
n, nk = sys.argv[1:3]
for row in csv.DictReader(open('/tmp/tr/t_kernel_stats.csv')):
    name = row['Name']
    if 'herm_tridiag' in name:
        print("n=%s nk=%s: %s %.3f ms (min %.3f)" % (n, nk, name.split('::')[1].split('(')[0], float(row['AverageNs']) / 1e6, float(row['MinNs']) / 1e6))
PY
done
