#!/bin/bash
# Round-3 refresh after the matrix-pipe Cholesky: driver-style bench, latency rows, secondary shapes, rocprof passes.
O=gpurun_out/r3c; mkdir -p $O
b() { tag=$1; shift; timeout 150 python bench.py "$@" > $O/$tag.json 2> $O/$tag.err; }
b bench_n1
for B in 1 8 64 256 1024; do
  b latency_b$B --batch $B --steps 200 --warmup 20 --no-cpu-baseline --no-mixed
  b latency_b${B}_noprof --batch $B --steps 200 --warmup 20 --no-cpu-baseline --no-mixed --no-profile
done
Q="--steps 10 --warmup 3 --no-cpu-baseline --no-mixed"
b config2_n150 --state-dim 150 --features 50 $Q
b n150_m80 --state-dim 150 --features 40 $Q
b tumvi_n203_m60 --state-dim 203 --features 30 --batch 8192 $Q
b n100_m40 --state-dim 100 --features 20 $Q
b glevel --batch 4096 --level G $Q
b glevel_ransac --batch 4096 --level G --ransac $Q
b config3_oos --batch 4096 --level G --oos 20 $Q
b config3_oos_b16384 --batch 16384 --level G --oos 20 $Q
b frame_rk4 --batch 4096 --level G --propagate-samples 16 $Q
b frame_pd --batch 4096 --level G --propagate-samples 16 --integrator PrinceDormand $Q
b config4_n400_fp64 --state-dim 400 --features 150 --batch 4096 --steps 8 --warmup 2 --no-cpu-baseline --no-mixed
b n848_m250 --state-dim 848 --features 125 --batch 1024 --steps 8 --warmup 2 --no-cpu-baseline --no-mixed
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3c/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        s=d.get("stage_ms_per_step",{})
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],4), "chol", round(s.get("chol_S",0),4), "solve", round(s.get("trsm_gain",0),4), d.get("parity_check",{}).get("ok"), d.get("roofline",{}).get("frac"))
    except Exception as e: print(f, "ERR", e)
PY
bash scripts/collect_profiles.sh r03b > $O/collect.log 2>&1
tail -3 $O/collect.log
