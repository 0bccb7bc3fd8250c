#!/bin/bash
# Round-5 evidence, all on ONE box (run through gpurun): default bench line, rocprofv3 kernel trace + PMC passes of the same
# command, the solve kernel's floor ablations (timing-only libraries), and two ranks sharing the GPU. Output: gpurun_out/r05/
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
timeout 900 python bench.py --full-out $O/bench_full.json > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.json
BENCH_ARGS="" bash scripts/collect_profiles.sh r05 > $O/collect.log 2>&1; tail -3 $O/collect.log
Q="--no-cpu-baseline --no-mixed --no-configs --no-dropin --no-parity-check --steps 8 --warmup 2"
for rep in 1 2; do
  timeout 300 python bench.py $Q > $O/floor_default_$rep.json 2> $O/floor_default_$rep.err
  for n in abl2 abl11; do
    XIVO_HIP_LIBRARY=$R/xivo_amd/csrc/build/abl/libxivo_hip_$n.so timeout 300 python bench.py $Q > $O/floor_${n}_$rep.json 2> $O/floor_${n}_$rep.err
  done
done
python - "$O" <<'PY'
import json,glob,sys,os
for f in sorted(glob.glob(sys.argv[1]+"/floor_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); s=d.get("stage_ms",{})
        print(os.path.basename(f), round(d["ms_per_step"],3), {k:round(v,3) for k,v in s.items()})
    except Exception as e: print(f,"ERR",open(f.replace(".json",".err")).read()[-300:])
PY
HIP_VISIBLE_DEVICES=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-dropin > $O/bench_2ranks_on_1gpu.json 2> $O/bench_2ranks.err
tail -c 400 $O/bench_2ranks_on_1gpu.json
