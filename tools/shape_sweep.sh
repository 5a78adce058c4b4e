#!/bin/bash
# Does the geometry cost model pick the fastest compiled geometry?  Auto choice (0x0) beside the
# candidates, linear and affine, for common read lengths (developer tool; run on the GPU box).
for shape in "36 100 2097152 8x6,8x8,16x4" "64 128 2097152 8x8,16x4,8x10" "75 250 1048576 8x10,8x12,16x8" \
             "100 300 1048576 8x16,16x8,16x10" "150 500 1048576 16x10,8x20,16x12,32x8" "250 600 524288 16x16,32x8,32x10" \
             "300 1000 262144 32x10,32x12,64x8" "500 1500 131072 64x8,32x16,64x12" ; do
  set -- $shape
  echo "# R=$1 F=$2 n=$3"
  python tools/geom_sweep.py --R $1 --F $2 --n $3 --iters 3 --geoms 0x0,$4 2>&1 | grep -v amdgpu.ids | cut -c1-64
  python tools/geom_sweep.py --R $1 --F $2 --n $3 --iters 3 --geoms 0x0,$4 --affine 1 2>&1 | grep -v amdgpu.ids | cut -c1-64
done
