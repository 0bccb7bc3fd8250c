#!/bin/bash
# ab_env.sh <out-dir> "<bench args>" "ENV1=1" "ENV2=1" ...: the default library against itself under environment knobs, back to
# back on ONE box, twice each in alternating order ("default" = no knob); prints value / ms per step / stage times / parity.
O=$1; A=$2; shift 2
mkdir -p $O
for rep in 1 2; do
  for n in default "$@"; do
    tag=$(echo "$n" | tr -c 'A-Za-z0-9_=\n' '_')
    if [ "$n" = default ]; then timeout 300 python bench.py --no-cpu-baseline --no-mixed --no-configs --no-dropin $A > $O/${tag}_$rep.json 2> $O/${tag}_$rep.err
    else env $n timeout 300 python bench.py --no-cpu-baseline --no-mixed --no-configs --no-dropin $A > $O/${tag}_$rep.json 2> $O/${tag}_$rep.err; fi
  done
done
python - "$O" <<'PY'
import json,glob,sys,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); s=d.get("stage_ms",d.get("stage_ms_per_step",{}))
        print(os.path.basename(f), round(d["value"]), round(d["ms_per_step"],3), {k:round(v,3) for k,v in s.items()}, (d.get("parity_check") or {}).get("ok"), (d.get("parity_last") or d.get("parity_check_last_timed_step") or {}).get("ok"))
    except Exception as e: print(f,"ERR",open(f.replace(".json",".err")).read()[-300:])
PY
