#!/bin/bash
# resource usage (VGPRs, spills) of the one-kernel update's instantiations; extra -D defines as arguments
cd "$(dirname "$0")/../xivo_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-result -Wno-unused-value "$@" -Rpass-analysis=kernel-resource-usage -c fused_update.hip -o /tmp/fused_update_chk.o 2>&1 | grep -E "error|Function Name|VGPRs:|VGPRs Spill|ScratchSize" | sed -e 's/.*remark: *//' | paste - - - - 2>/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-result -Wno-unused-value "$@" -Rpass-analysis=kernel-resource-usage -c fused_update7.hip -o /tmp/fused_update_chk.o 2>&1 | grep -E "error|Function Name|VGPRs:|VGPRs Spill|ScratchSize" | sed -e 's/.*remark: *//' | paste - - - - 2>/dev/null
