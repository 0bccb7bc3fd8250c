#!/bin/bash
# A/B inside one box on the metric point, config 4 and the whole frame: default library against build/abl/libxivo_hip_$1.so
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
Q="--no-cpu-baseline --no-mixed --no-configs --no-dropin --steps 8 --warmup 3"
for rep in 1 2; do
for lib in default $@; do
  if [ $lib = default ]; then unset XIVO_HIP_LIBRARY; else export XIVO_HIP_LIBRARY=$R/xivo_amd/csrc/build/abl/libxivo_hip_$lib.so; fi
  for cfg in "" "--state-dim 400 --features 150 --batch 4096" "--level G --propagate-samples 16 --integrator PrinceDormand --batch 4096"; do
    python bench.py $Q $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$cfg', round(d['value']), round(d['ms_per_step'],4), d['parity_check']['ok'], {k:round(v,3) for k,v in d['stage_ms'].items()})"
  done
done
done
