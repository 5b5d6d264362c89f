#!/bin/bash
# One round's rocprofv3 evidence on the GPU box (run through gpurun from the repository root):
#   bash tools/profile_round.sh r04
# kernel trace + stats of the default bench line and of cfg3 / cfg4 / cfg5, and the PMC passes of
# MI355X_MICROARCH.md (one counter group per pass, --kernel-trace only beside --pmc, the program itself after `--`)
# for cfg2, cfg3, cfg4 and cfg5, each with one warm-up step in front of the measured one (the summariser drops the
# warm-up step's launches: the first half of every kernel's dispatches).
# Output under gpurun_out/<tag>/; `python tools/summarize_profiles.py <tag> gpurun_out/<tag>` turns it into profiles/.
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$TAG
mkdir -p $O
B="python3 $R/bench.py --cpu-sample 0"
export TBK_BENCH_SKIP_PEAK=1
export TBK_BENCH_SKIP_CONFIGS=1   # the line's "configs" runs would put other shapes into the same kernel names
export TBK_BENCH_SKIP_HOSTAPI=1   # ... and the one-k calls of host_api / single_k_us many small launches: full-size launches only
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- $B > $O/trace.log 2>&1
for c in cfg1 cfg3 cfg4 cfg5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$c -o bench -- $B --config $c --steps 2 > $O/trace_$c.log 2>&1
done
for c in cfg2 cfg3 cfg4 cfg5; do
  P="$B --config $c --steps 1 --warmup 1"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${c}_fetch -o bench -- $P > $O/${c}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${c}_write -o bench -- $P > $O/${c}_write.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/${c}_sq -o bench -- $P > $O/${c}_sq.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/${c}_sq2 -o bench -- $P > $O/${c}_sq2.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/${c}_tcc -o bench -- $P > $O/${c}_tcc.log 2>&1
done
tail -2 $O/trace.log
find $O -name "*stats*" | head
