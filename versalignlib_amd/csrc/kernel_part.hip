// kernel_part.hip -- definitions of the per-geometry kernels of part VALIGN_PART (see
// kernel_instances.hip.h); compiled once per part, linked into libHIPKernel.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#define VALIGN_KERNEL_PART_TU 1
#include "kernel_instances.hip.h"

#ifndef VALIGN_PART
#error "compile with -DVALIGN_PART=<0..VALIGN_KERNEL_PARTS-1>"
#endif

namespace valign {

#define VALIGN_DEFINE(G, K) VALIGN_GEOMETRY_KERNELS(template, G, K)
#if VALIGN_PART == 0
VALIGN_GEOMETRIES_PART0(VALIGN_DEFINE)
#elif VALIGN_PART == 1
VALIGN_GEOMETRIES_PART1(VALIGN_DEFINE)
#elif VALIGN_PART == 2
VALIGN_GEOMETRIES_PART2(VALIGN_DEFINE)
#elif VALIGN_PART == 3
VALIGN_GEOMETRIES_PART3(VALIGN_DEFINE)
#elif VALIGN_PART == 4
VALIGN_GEOMETRIES_PART4(VALIGN_DEFINE)
#elif VALIGN_PART == 5
VALIGN_GEOMETRIES_PART5(VALIGN_DEFINE)
#elif VALIGN_PART == 6
VALIGN_GEOMETRIES_PART6(VALIGN_DEFINE)
#else
#error "VALIGN_PART out of range"
#endif

}  // namespace valign
