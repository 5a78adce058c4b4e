#!/bin/bash
# tools/copy_contention_probe.py under the runtime's copy-engine switches (developer tool, run on the GPU box)
python tools/copy_contention_probe.py 2>&1 | grep -v amdgpu.ids
echo "== HSA_ENABLE_SDMA=0"; HSA_ENABLE_SDMA=0 python tools/copy_contention_probe.py 2>&1 | grep -v amdgpu.ids
echo "== GPU_FORCE_BLIT_COPY_SIZE=0"; GPU_FORCE_BLIT_COPY_SIZE=0 python tools/copy_contention_probe.py 2>&1 | grep -v amdgpu.ids
