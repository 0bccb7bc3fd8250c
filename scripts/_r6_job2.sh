#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fused_gpu.py -q --tb=line 2>&1 | tail -40 > gpurun_out/fused_tests.log
cat gpurun_out/fused_tests.log
timeout 300 python bench.py --sub --no-cpu-baseline --state-dim 203 --features 30 --batch 8192 --steps 20 --warmup 3 > gpurun_out/ab_203.json 2> gpurun_out/ab_203.err; tail -c 600 gpurun_out/ab_203.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/ab_203.json").read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],3), d["roofline"]["kernel"], round(d["roofline"]["avg_launch_ms"],3), d["parity_check"]["ok"], d["parity_check_last_timed_step"].get("ok"))
PY
