#!/bin/bash
# Builds one library per timing-only ablation of the in-solve covariance update (XIVO_ABL in chol_trsm.hip) into
# xivo_amd/csrc/build/abl/libxivo_hip_abl<N>.so; time them on the GPU box with XIVO_HIP_LIBRARY=... python bench.py --no-parity-check
cd "$(dirname "$0")/../xivo_amd/csrc"
mkdir -p build/abl
for n in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-result -DXIVO_ABL=$n -c chol_trsm.hip -o build/abl/chol_trsm_$n.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libxivo_hip_abl$n.so build/gemm_f64.o build/gemm_sym_f64.o build/chol_f64.o build/abl/chol_trsm_$n.o build/ekf_kernels.o build/ell_kernels.o build/ldlt_fallback.o build/capi.o && echo built $n ) &
done
wait
