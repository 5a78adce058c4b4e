#!/bin/bash
# align_device on 1 M pairs (linear + affine, SW + NW): the 7/8 + 1/8 split against geometric parts (developer tool)
for parts in 2 3 4 5; do
  echo "== VALIGN_HIP_SPLIT_PARTS=$parts"
  VALIGN_HIP_SPLIT_PARTS=$parts python tools/align_bench.py --iters 3 2>&1 | grep -v amdgpu.ids
done
