R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export TBK_BENCH_SKIP_PEAK=1 TBK_BENCH_SKIP_CONFIGS=1
O=$R/gpurun_out/r03/csr
mkdir -p $O
P="python3 $R/bench.py --cpu-sample 0 --config cfg3 --steps 1 --warmup 1 "
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/sq -o b -- $P > $O/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/sq2 -o b -- $P > $O/sq2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/tcc -o b -- $P > $O/tcc.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/r03/csr"
for g in ("sq","sq2","tcc","fw"):
    agg=collections.defaultdict(list)
    for f in glob.glob(O+"/"+g+"/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if "hk_csr" in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,v in sorted(agg.items()):
        print(g, k, "%.4g"%(sum(v[len(v)//2:])/max(1,len(v[len(v)//2:]))), len(v))
PY
