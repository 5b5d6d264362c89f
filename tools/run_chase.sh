# measurement helper (GPU box): per-kernel times of the two-stage reduction for (n, batch, chase waves, chase stagger)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "$@"; do
  set -- $cfg
  export TBK_CHASE_NW=$3 TBK_CHASE_STAGGER=$4
  rm -rf /tmp/tr; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o t -- $R/tools/band_time $1 $2 > /dev/null 2>&1
  python3 - "$1" "$2" "$3" "$4" <<'PY'
import csv, sys
n, nk, nw, st = sys.argv[1:5]
out = []
for row in csv.DictReader(open('/tmp/tr/t_kernel_stats.csv')):
    name = row['Name']
    if 'band_' in name:
        out.append("%s %.3f us/matrix" % (name.split('::')[1].split('(')[0], float(row['AverageNs']) / 1000 / int(nk)))
print("n=%s nk=%s NW=%s stagger=%s: %s" % (n, nk, nw, st, "; ".join(out)))
PY
done
