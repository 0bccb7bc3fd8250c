#!/bin/bash
# A/B of the Cholesky instantiations on ONE box (nodes of the pool differ): default library (k-slice loop inside
# factor_invert_diag) against a build with all sixteen columns unrolled (-DXIVO_CHOL_UNROLL16=1 -> libxivo_hip_u16.so).
O=gpurun_out/s3; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 120 python bench.py $ARGS --no-cpu-baseline --no-mixed > $O/$tag.json 2> $O/$tag.err; }
( cd xivo_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-result -Wno-unused-value -DXIVO_CHOL_UNROLL16=1 -c chol_f64.hip -o build/chol_f64_u16.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libxivo_hip_u16.so build/gemm_f64.o build/gemm_sym_f64.o build/chol_f64_u16.o build/chol_trsm.o build/ekf_kernels.o build/ell_kernels.o build/ldlt_fallback.o build/capi.o ) 2>/dev/null
U16=XIVO_HIP_LIBRARY=$PWD/xivo_amd/libxivo_hip_u16.so
ARGS="--steps 6 --warmup 2"
run a_loop A=1
run a_u16 $U16
run a_loop_pre XIVO_HIP_CHOL_LOOKAHEAD=1
run a_u16_pre $U16 XIVO_HIP_CHOL_LOOKAHEAD=1
run a_wave XIVO_HIP_CHOL_WAVE=1
run a_loop2 A=1
run a_u16_2 $U16
run a_minb2 XIVO_HIP_CHOL_MINB2=1
ARGS="--batch 1 --steps 200 --warmup 20"
run b1_loop A=1
run b1_u16 $U16
ARGS="--batch 64 --steps 100 --warmup 10"
run b64_loop A=1
run b64_u16 $U16
ARGS="--batch 1024 --steps 50 --warmup 10"
run b1024_loop A=1
run b1024_u16 $U16
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s3/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        s=d.get("stage_ms",d.get("stage_ms_per_step",{}))
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],4), "chol", round(s.get("chol_S",0),4), d.get("parity_check",{}).get("ok"))
    except Exception as e: print(f, "ERR", e)
PY
