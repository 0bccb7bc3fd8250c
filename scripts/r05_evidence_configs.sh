#!/bin/bash
# Kernel trace + PMC passes (scripts/collect_profiles.sh) of three more rows of the bench line's `configs`, one box.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
Q="--no-mixed --no-configs --no-dropin"
BENCH_ARGS="$Q --state-dim 150 --features 50 --batch 16384" bash scripts/collect_profiles.sh r05cfg2 > /dev/null 2>&1
BENCH_ARGS="$Q --state-dim 400 --features 150 --batch 4096 --flags 16384 --tol-P 5e-5 --tol-dx-last 1e-2" bash scripts/collect_profiles.sh r05cfg4f32w > /dev/null 2>&1
BENCH_ARGS="$Q --level G --calib --batch 4096" bash scripts/collect_profiles.sh r05calib > /dev/null 2>&1
for t in r05cfg2 r05cfg4f32w r05calib; do ls gpurun_out/prof_$t/summary; head -6 gpurun_out/prof_$t/summary/${t}_kernel_stats.csv; done
