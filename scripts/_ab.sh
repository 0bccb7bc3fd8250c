python -m pytest tests -q -m gpu 2>&1 | tail -4
python bench.py --steps 10 --warmup 3 > gpurun_out/r02_bench_n1.json 2> gpurun_out/r02_bench_n1.err
python bench.py --level G --batch 4096 --no-cpu-baseline > gpurun_out/r02_bench_glevel.json 2>/dev/null
python bench.py --level G --batch 4096 --ransac --no-cpu-baseline > gpurun_out/r02_bench_glevel_ransac.json 2>/dev/null
python bench.py --level G --batch 4096 --propagate-samples 16 --no-cpu-baseline > gpurun_out/r02_bench_frame_rk4.json 2>/dev/null
python bench.py --level G --batch 4096 --propagate-samples 16 --integrator PrinceDormand --no-cpu-baseline > gpurun_out/r02_bench_frame_pd.json 2>/dev/null
python bench.py --state-dim 150 --features 50 --no-cpu-baseline > gpurun_out/r02_bench_config2_n150.json 2>gpurun_out/c2.err
