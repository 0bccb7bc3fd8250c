#!/bin/bash
# build_variant.sh <name> "<-D defines>" [file.hip ...]: an A/B build of the library with extra defines on the named kernel
# files (default chol_trsm.hip) -> xivo_amd/csrc/build/abl/libxivo_hip_<name>.so; run with XIVO_HIP_LIBRARY=<that file>
# (node-to-node variance exceeds most kernel deltas: variants are timed against the default inside ONE gpurun call).
cd "$(dirname "$0")/../xivo_amd/csrc"
name=$1; defs=$2; shift 2
files=${@:-chol_trsm.hip}
# (fused_update7.hip includes fused_update.hip: a variant of the one is a variant of the other)
if [[ " $files " == *" fused_update.hip "* ]]; then files="$files fused_update7.hip"; fi
mkdir -p build/abl
objs=""
for f in gemm_f64 gemm_sym_f64 chol_f64 chol_trsm solve_fused fused_update fused_update7 ekf_kernels ell_kernels ldlt_fallback dropin capi; do
  if [[ " $files " == *" $f.hip "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-result -Wno-unused-value $defs -c $f.hip -o build/abl/${f}_$name.o || exit 1
    objs="$objs build/abl/${f}_$name.o"
  else
    objs="$objs build/$f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libxivo_hip_$name.so $objs && echo built $name
