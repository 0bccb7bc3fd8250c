#!/bin/bash
# ab_bench.sh <out-dir> "<bench args>" name1 name2 ...: the default library and the named variants (scripts/build_variant.sh)
# timed back to back on ONE box, twice each in alternating order; prints value / ms per step / solve-stage ms / parity.
O=$1; A=$2; shift 2
mkdir -p $O
L=xivo_amd/csrc/build/abl
for rep in 1 2; do
  for n in default "$@"; do
    if [ $n = default ]; then unset XIVO_HIP_LIBRARY; else export XIVO_HIP_LIBRARY=$PWD/$L/libxivo_hip_$n.so; fi
    timeout 300 python bench.py --no-cpu-baseline --no-mixed --no-configs --no-dropin $A > $O/${n}_$rep.json 2> $O/${n}_$rep.err
  done
done
unset XIVO_HIP_LIBRARY
python - "$O" <<'PY'
import json,glob,sys,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); s=d.get("stage_ms",d.get("stage_ms_per_step",{}))
        print(os.path.basename(f), round(d["value"]), round(d["ms_per_step"],3), {k:round(v,3) for k,v in s.items()}, (d.get("parity_check") or {}).get("ok"), (d.get("parity_last") or d.get("parity_check_last_timed_step") or {}).get("ok"))
    except Exception as e: print(f,"ERR",open(f.replace(".json",".err")).read()[-300:])
PY
