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

#define VALIGN_DEFINE_FULL(G, K) VALIGN_FAST_KERNELS(template, G, K) VALIGN_FALLBACK_KERNELS(template, G, K)
#define VALIGN_DEFINE_FAST(G, K) VALIGN_FAST_KERNELS(template, G, K)
#if VALIGN_PART == 0
VALIGN_PART0(VALIGN_DEFINE_FULL, VALIGN_DEFINE_FAST)
#elif VALIGN_PART == 1
VALIGN_PART1(VALIGN_DEFINE_FULL, VALIGN_DEFINE_FAST)
#elif VALIGN_PART == 2
VALIGN_PART2(VALIGN_DEFINE_FULL, VALIGN_DEFINE_FAST)
#elif VALIGN_PART == 3
VALIGN_PART3(VALIGN_DEFINE_FULL, VALIGN_DEFINE_FAST)
#elif VALIGN_PART == 4
VALIGN_PART4(VALIGN_DEFINE_FULL, VALIGN_DEFINE_FAST)
#elif VALIGN_PART == 5
VALIGN_PART5(VALIGN_DEFINE_FULL, VALIGN_DEFINE_FAST)
#else
#error "VALIGN_PART out of range"
#endif

}  // namespace valign
