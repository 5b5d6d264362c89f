#!/bin/bash
# One round's rocprofv3 evidence on the GPU box (run through gpurun from the repository root):
#   kernel trace + stats of the default bench line and of cfg3 / cfg4 / cfg5, and the PMC passes of
#   MI355X_MICROARCH.md (one counter group per pass, --kernel-trace only beside --pmc).
# Output under gpurun_out/r01n/; `python tools/summarize_profiles.py r01 gpurun_out/r01n` turns it into profiles/.
set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r01n
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 $R/bench.py --cpu-sample 0 > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o bench -- python3 $R/bench.py --cpu-sample 0 --steps 1 --warmup 0 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o bench -- python3 $R/bench.py --cpu-sample 0 --steps 1 --warmup 0 > $O/write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/sq -o bench -- python3 $R/bench.py --cpu-sample 0 --steps 1 --warmup 0 > $O/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/sq2 -o bench -- python3 $R/bench.py --cpu-sample 0 --steps 1 --warmup 0 > $O/sq2.log 2>&1
for c in cfg3 cfg4 cfg5; do rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$c -o bench -- python3 $R/bench.py --cpu-sample 0 --config $c --steps 2 > $O/trace_$c.log 2>&1; done
tail -2 $O/trace.log
find $O -name "*stats*" | head
