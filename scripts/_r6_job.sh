#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fused_gpu.py -q 2>&1 | tail -30 > gpurun_out/fused_tests.log
timeout 300 python scripts/trace_fused.py 203 30 8192 > gpurun_out/ftrace_203.json 2>&1
timeout 300 python scripts/trace_fused.py 150 50 16384 > gpurun_out/ftrace_150.json 2>&1
L=$PWD/xivo_amd/csrc/build/abl
for cfg in "203 30 8192" "150 50 16384"; do
  set -- $cfg
  for n in default fabl1; do
    if [ $n = default ]; then unset XIVO_HIP_LIBRARY; P=""; else export XIVO_HIP_LIBRARY=$L/libxivo_hip_$n.so; P="--no-parity-check --no-last-step-parity"; fi
    timeout 300 python bench.py --sub --no-cpu-baseline $P --state-dim $1 --features $2 --batch $3 --steps 20 --warmup 3 > gpurun_out/ab_$1_$n.json 2> gpurun_out/ab_$1_$n.err
  done
  unset XIVO_HIP_LIBRARY
done
python - <<'PY'
import json
for f in ["ftrace_203","ftrace_150"]:
    try:
        txt=open(f"gpurun_out/{f}.json").read(); d=json.loads(txt[:txt.index("per wave")]); print(f, round(d["total_us_median"]*100), {k[:12]:round(v*100) for k,v in d["segments_us"].items()})
    except Exception as e: print(f, "ERR", e, open(f"gpurun_out/{f}.json").read()[-800:])
for s in ("203","150"):
  for n in ["default","fabl1"]:
    f=f"ab_{s}_{n}"
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"]), round(d["ms_per_step"],3), d["roofline"]["kernel"], round(d["roofline"]["avg_launch_ms"],3), (d.get("parity_check") or {}).get("ok"), (d.get("parity_check_last_timed_step") or {}).get("ok"))
    except Exception as e: print(f, "ERR", e, open(f"gpurun_out/{f}.err").read()[-300:])
PY
grep -A14 "per wave: gather" gpurun_out/ftrace_203.json
grep -E "FAILED|passed|failed|Error" gpurun_out/fused_tests.log | head -20
