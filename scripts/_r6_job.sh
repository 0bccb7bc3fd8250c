#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fused_gpu.py -q --tb=line 2>&1 | tail -25 > gpurun_out/fused_tests.log
timeout 300 python scripts/trace_fused.py 203 30 8192 > gpurun_out/ftrace_203.json 2>&1
timeout 300 python scripts/trace_fused.py 150 50 16384 > gpurun_out/ftrace_150.json 2>&1
for cfg in "203 30 8192" "150 50 16384"; do
  set -- $cfg
  timeout 300 python bench.py --sub --no-cpu-baseline --state-dim $1 --features $2 --batch $3 --steps 20 --warmup 3 > gpurun_out/ab_$1_default.json 2> gpurun_out/ab_$1_default.err
done
python - <<'PY'
import json
for f in ["ftrace_203","ftrace_150"]:
    try:
        txt=open(f"gpurun_out/{f}.json").read(); d=json.loads(txt[:txt.index("per wave")]); print(f, round(d["total_us_median"]*100), {k[:12]:round(v*100) for k,v in d["segments_us"].items()}, {k[:10]:round(v*100) for k,v in d["phase2_us"].items()})
    except Exception as e: print(f, "ERR", e, open(f"gpurun_out/{f}.json").read()[-800:])
for s in ("203","150"):
    f=f"ab_{s}_default"
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"]), round(d["ms_per_step"],3), d["roofline"]["kernel"], round(d["roofline"]["avg_launch_ms"],3), d["parity_check"]["ok"], d["parity_check"]["rel_fro_P_max"], d["parity_check"]["rel_dx_max"], (d.get("parity_check_last_timed_step") or {}).get("ok"))
    except Exception as e: print(f, "ERR", e, open(f"gpurun_out/{f}.err").read()[-600:])
PY
grep -E "^/root|FAILED|passed|failed|Error" gpurun_out/fused_tests.log | head -20
