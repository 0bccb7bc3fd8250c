#!/bin/bash
# Round-6 evidence, all on ONE box (run through gpurun): the default bench line, rocprofv3 kernel trace + PMC passes of the
# headline command and of the TUM-VI / config-2 / whole-frame / config-4 rows (scripts/collect_profiles.sh, each counter
# group in its own run), 2 and 8 ranks sharing the one GPU, and the point-cloud-world sequences. Output: gpurun_out/r06/
# and gpurun_out/prof_r06*/summary/; copy what is to be judged into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
PART=${1:-all}
Q="--no-mixed --no-configs --no-dropin"
if [ $PART = all ] || [ $PART = prof ]; then
BENCH_ARGS="" bash scripts/collect_profiles.sh r06 > $O/collect.log 2>&1; tail -2 $O/collect.log
BENCH_ARGS="$Q --state-dim 203 --features 30 --batch 8192" bash scripts/collect_profiles.sh r06tumvi > /dev/null 2>&1
BENCH_ARGS="$Q --state-dim 150 --features 50 --batch 16384" bash scripts/collect_profiles.sh r06cfg2 > /dev/null 2>&1
BENCH_ARGS="$Q --level G --propagate-samples 16 --integrator PrinceDormand --batch 4096" bash scripts/collect_profiles.sh r06frame > /dev/null 2>&1
BENCH_ARGS="$Q --state-dim 400 --features 150 --batch 4096" bash scripts/collect_profiles.sh r06cfg4 > /dev/null 2>&1
for t in r06 r06tumvi r06cfg2 r06frame r06cfg4; do head -4 gpurun_out/prof_$t/summary/${t}_kernel_stats.csv; done
fi
if [ $PART = all ] || [ $PART = bench ]; then
timeout 900 python bench.py --full-out $O/bench_full.json > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.json
for n in 2 8; do
# (eight contexts of 16384 filters do not fit 288 GB: 4096 filters per rank there)
B=16384; [ $n = 8 ] && B=4096
HIP_VISIBLE_DEVICES=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-dropin > $O/bench_${n}ranks_on_1gpu.json 2> $O/bench_${n}ranks.err
tail -c 300 $O/bench_${n}ranks_on_1gpu.json
done
fi
if [ $PART = all ] || [ $PART = pcw ]; then
timeout 900 python scripts/run_pcw.py -vectorized -sequences 4096 > $O/pcw_4096.json 2> $O/pcw_4096.err; tail -c 600 $O/pcw_4096.json
timeout 900 python scripts/run_pcw.py -vectorized -sequences 512 > $O/pcw_512.json 2> $O/pcw_512.err; tail -c 600 $O/pcw_512.json
timeout 900 python scripts/run_pcw.py -host cpp -sequences 1 > $O/pcw_1.json 2> $O/pcw_1.err; tail -c 600 $O/pcw_1.json
fi
