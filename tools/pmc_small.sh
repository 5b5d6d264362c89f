# PMC passes of the n <= 64 reduction / QL kernels on tools/race_check.py (GPU box):  bash tools/pmc_small.sh <n> <batch> <out dir>
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=$1; NK=$2; O=$R/$3
mkdir -p $O
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $O/sq -o t -- python3 $R/tools/race_check.py $N $NK 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/sq2 -o t -- python3 $R/tools/race_check.py $N $NK 3 > /dev/null 2>&1
python3 - $O $NK <<'PY'
import csv, glob, sys, collections
o, nk = sys.argv[1], int(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(o + '/*/t_counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('::')[-1].split('(')[0]
        agg[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k, cs in agg.items():
    if 'tridiag' not in k: continue
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-32s %.4g per launch (%d launches)  %.4g per matrix" % (c, sum(v) / len(v), len(v), sum(v) / len(v) / nk))
PY
