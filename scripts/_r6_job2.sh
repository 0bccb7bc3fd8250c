#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/gpu_tests_all.log
grep -E "^FAILED|passed|failed" gpurun_out/gpu_tests_all.log
