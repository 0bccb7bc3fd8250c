#!/bin/bash
# timing-only ablation libraries of the one-kernel update against the default, one box (results of the ablations are wrong by design)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
Q="--no-cpu-baseline --no-mixed --no-configs --no-dropin --no-parity-check --steps 10 --warmup 3"
for lib in default $@; do
  if [ $lib = default ]; then unset XIVO_HIP_LIBRARY; else export XIVO_HIP_LIBRARY=$R/xivo_amd/csrc/build/abl/libxivo_hip_$lib.so; fi
  for cfg in "--state-dim 203 --features 30 --batch 8192" "--state-dim 150 --features 50 --batch 16384"; do
    python bench.py $Q $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$cfg', round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['stage_ms'].items()})"
  done
done
