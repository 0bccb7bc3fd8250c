#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q --deselect tests/test_bench_gpu.py 2>&1 | tail -60 > gpurun_out/gpu_tests_all.log
grep -E "^FAILED|passed|failed" gpurun_out/gpu_tests_all.log
timeout 900 python -m pytest tests/test_bench_gpu.py -m gpu -q 2>&1 | tail -30 > gpurun_out/gpu_tests_bench.log
grep -E "^FAILED|passed|failed" gpurun_out/gpu_tests_bench.log
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 3000 gpurun_out/bench_default.json
