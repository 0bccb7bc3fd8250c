python -m pytest tests/test_update_gpu.py tests/test_bench_gpu.py -q -m gpu 2>&1 | tail -3
bash scripts/collect_profiles.sh r02 2>&1 | tail -2
cp gpurun_out/prof_r02/summary/r02_pmc_summary.json profiles/r02_pmc_summary.json
python bench.py --steps 10 --warmup 3 > gpurun_out/r02_bench_n1.json 2> gpurun_out/r02_bench_n1.err
