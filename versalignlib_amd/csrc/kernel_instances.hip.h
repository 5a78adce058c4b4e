// kernel_instances.hip.h -- which (G, K) geometries exist and in which translation unit their 22
// kernels are compiled.  The plugin is one shared object, but 340 kernel instances in one hipcc run
// take minutes: hip_plugin.hip only declares them (extern template), kernel_part.hip is compiled
// once per part (-DVALIGN_PART=n, in parallel, versalignlib_amd/build.py) and defines them.
#pragma once

#include "dp_kernels.hip.h"
#include "trace_kernels.hip.h"

// Parts are balanced by rows per lane (compile time grows with K).
#define VALIGN_GEOMETRIES_PART0(X) X(64, 32) X(8, 4) X(8, 6)
#define VALIGN_GEOMETRIES_PART1(X) X(64, 24) X(8, 12) X(16, 4)
#define VALIGN_GEOMETRIES_PART2(X) X(8, 20) X(8, 16) X(8, 8)
#define VALIGN_GEOMETRIES_PART3(X) X(16, 16) X(32, 16) X(8, 10)
#define VALIGN_GEOMETRIES_PART4(X) X(64, 16) X(16, 12) X(32, 10)
#define VALIGN_GEOMETRIES_PART5(X) X(32, 12) X(64, 12) X(16, 10) X(64, 8)
#define VALIGN_GEOMETRIES_PART6(X) X(16, 8) X(32, 8)
#define VALIGN_KERNEL_PARTS 7

#define VALIGN_ALL_GEOMETRIES(X)                                                                          \
    VALIGN_GEOMETRIES_PART0(X) VALIGN_GEOMETRIES_PART1(X) VALIGN_GEOMETRIES_PART2(X) VALIGN_GEOMETRIES_PART3(X) \
    VALIGN_GEOMETRIES_PART4(X) VALIGN_GEOMETRIES_PART5(X) VALIGN_GEOMETRIES_PART6(X)

// Every kernel of one geometry; PREFIX is `extern template` (declaration) or `template` (definition).
#define VALIGN_GEOMETRY_KERNELS(PREFIX, G, K)                                                        \
    PREFIX __global__ void score_kernel<G, K, kAlgSW, kGapLinear>(const ScoreArgs);                  \
    PREFIX __global__ void score_kernel<G, K, kAlgSW, kGapSym>(const ScoreArgs);                     \
    PREFIX __global__ void score_kernel<G, K, kAlgSW, kGapAffine>(const ScoreArgs);                  \
    PREFIX __global__ void score_kernel<G, K, kAlgSW, kGapAffineSym>(const ScoreArgs);               \
    PREFIX __global__ void score_kernel<G, K, kAlgSW, kGapAffineSymF16>(const ScoreArgs);            \
    PREFIX __global__ void score_kernel<G, K, kAlgSW, kGapAffineF16>(const ScoreArgs);               \
    PREFIX __global__ void score_kernel<G, K, kAlgNW, kGapLinear>(const ScoreArgs);                  \
    PREFIX __global__ void score_kernel<G, K, kAlgNW, kGapSym>(const ScoreArgs);                     \
    PREFIX __global__ void score_kernel<G, K, kAlgNW, kGapAffine>(const ScoreArgs);                  \
    PREFIX __global__ void score_kernel<G, K, kAlgNW, kGapAffineSym>(const ScoreArgs);               \
    PREFIX __global__ void score_kernel<G, K, kAlgNW, kGapAffineSymF16>(const ScoreArgs);            \
    PREFIX __global__ void score_kernel<G, K, kAlgNW, kGapAffineF16>(const ScoreArgs);               \
    PREFIX __global__ void score_kernel<G, K, kAlgNW, kGapSymF16>(const ScoreArgs);                  \
    PREFIX __global__ void score_kernel<G, K, kAlgSW, kGapSymF16>(const ScoreArgs);                  \
    PREFIX __global__ void align_fill_kernel<G, K, kAlgSW, false>(const FillArgs);                   \
    PREFIX __global__ void align_fill_kernel<G, K, kAlgSW, true>(const FillArgs);                    \
    PREFIX __global__ void align_fill_kernel<G, K, kAlgNW, false>(const FillArgs);                   \
    PREFIX __global__ void align_fill_kernel<G, K, kAlgNW, true>(const FillArgs);                    \
    PREFIX __global__ void align_fill_affine_kernel<G, K, kAlgSW, false>(const FillArgs);            \
    PREFIX __global__ void align_fill_affine_kernel<G, K, kAlgNW, false>(const FillArgs);            \
    PREFIX __global__ void align_fill_affine_kernel<G, K, kAlgSW, true>(const FillArgs);             \
    PREFIX __global__ void align_fill_affine_kernel<G, K, kAlgNW, true>(const FillArgs);             \
    PREFIX __global__ void align_fill_sse_kernel<G, K, kAlgSW>(const FillArgs);                      \
    PREFIX __global__ void align_fill_sse_kernel<G, K, kAlgNW>(const FillArgs);                      \
    PREFIX __global__ void align_fill_affine_tag_kernel<G, K, kAlgSW, false>(const FillArgs);        \
    PREFIX __global__ void align_fill_affine_tag_kernel<G, K, kAlgSW, true>(const FillArgs);         \
    PREFIX __global__ void align_fill_affine_tag_kernel<G, K, kAlgNW, false>(const FillArgs);        \
    PREFIX __global__ void align_fill_affine_tag_kernel<G, K, kAlgNW, true>(const FillArgs);         \
    PREFIX __global__ void align_fill_tag_kernel<G, K, kAlgSW, false, false>(const FillArgs);        \
    PREFIX __global__ void align_fill_tag_kernel<G, K, kAlgSW, true, false>(const FillArgs);         \
    PREFIX __global__ void align_fill_tag_kernel<G, K, kAlgSW, true, false, false, true>(const FillArgs); \
    PREFIX __global__ void align_fill_tag_kernel<G, K, kAlgNW, false, false>(const FillArgs);        \
    PREFIX __global__ void align_fill_tag_kernel<G, K, kAlgSW, false, true>(const FillArgs);         \
    PREFIX __global__ void align_fill_tag_kernel<G, K, kAlgSW, true, true>(const FillArgs);          \
    PREFIX __global__ void align_fill_tag_kernel<G, K, kAlgNW, false, true>(const FillArgs);
