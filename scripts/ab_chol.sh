mkdir -p gpurun_out/s2
run() { tag=$1; shift; env "$@" timeout 120 python bench.py $ARGS --no-cpu-baseline --no-mixed > gpurun_out/s2/$tag.json 2> gpurun_out/s2/$tag.err; }
ARGS="--steps 6 --warmup 2"
run wave XIVO_HIP_CHOL_WAVE=1
run reg3 XIVO_HIP_CHOL_REG=1
run reg3_nopre XIVO_HIP_CHOL_REG=1 XIVO_HIP_CHOL_NO_LOOKAHEAD=1
run reg2_up XIVO_HIP_CHOL_REG=1 XIVO_HIP_CHOL_MINB2=1
run reg2_lazy XIVO_HIP_CHOL_REG=1 XIVO_HIP_CHOL_MINB2=1 XIVO_HIP_CHOL_LAZY_LOADS=1
ARGS="--batch 1 --steps 200 --warmup 20"
run lat_up A=1
run lat_lazy XIVO_HIP_CHOL_LAZY_LOADS=1
ARGS="--batch 64 --steps 100 --warmup 10"
run lat64_up A=1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s2/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        s=d.get("stage_ms_per_step",{})
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],4), "chol", round(s.get("chol_S",0),4), d.get("parity_check",{}).get("ok"))
    except Exception as e: print(f, "ERR", e)
PY
