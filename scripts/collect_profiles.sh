#!/bin/bash
# Run on the GPU box (e.g. through gpurun): rocprofv3 kernel trace + PMC passes of the bench command,
# each counter group in its own run (never combined with sys/hip traces). Output: gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile ${BENCH_ARGS:-}"
echo "$BENCH" > $O/command.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o stats --output-format csv -- $BENCH > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O -o mfma --output-format csv -- $BENCH > $O/mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o fetch --output-format csv -- $BENCH > $O/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o write --output-format csv -- $BENCH > $O/write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O -o l2 --output-format csv -- $BENCH > $O/l2.log 2>&1
cd $R
python scripts/summarize_profiles.py $O $TAG
ls $O
