#!/bin/bash
# round 6, first GPU pass of the one-kernel update: parity tests, then A/B against the multi-kernel pipeline
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fused_gpu.py -x -q 2>&1 | tail -30 > gpurun_out/fused_tests.log
for cfg in "203 30 8192" "150 50 16384"; do
  set -- $cfg
  for fl in 0 65536; do
    timeout 300 python bench.py --sub --no-cpu-baseline --state-dim $1 --features $2 --batch $3 --flags $fl --steps 20 --warmup 3 > gpurun_out/ab_$1_$2_$fl.json 2> gpurun_out/ab_$1_$2_$fl.err
  done
done
tail -5 gpurun_out/fused_tests.log
for f in gpurun_out/ab_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d.get("value"), d.get("ms_per_step"), d.get("stage_ms"), d.get("parity_check"))
except Exception as e:
    print("ERR", e, open(sys.argv[1]).read()[-500:])
PY
done
