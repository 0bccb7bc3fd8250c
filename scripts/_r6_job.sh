#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for cfg in "203 30" "250 80" "150 50"; do
  set -- $cfg
  timeout 300 python bench.py --sub --no-cpu-baseline --state-dim $1 --features $2 --batch 1 --steps 200 --warmup 20 > gpurun_out/b1_$1.json 2> gpurun_out/b1_$1.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/b1_$1.json").read().strip().splitlines()[-1])
print("$1", round(d["ms_per_step"],4), {k: round(v,4) for k,v in d["stage_ms_per_step"].items()}, d["parity_check"]["ok"])
PY
done
