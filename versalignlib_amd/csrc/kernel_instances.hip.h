// kernel_instances.hip.h -- which (G, K) geometries exist, which kernels each one carries, and in which translation
// unit they are compiled.  The plugin is one shared object, but several hundred kernel instances in one hipcc run take
// minutes: the engine units only declare them (extern template), kernel_part.hip is compiled once per part
// (-DVALIGN_PART=n, in parallel, versalignlib_amd/build.py) and defines them.
//
// Round 4 pruned the table to what the engine can select (21 geometries x 35 kernels = 735 instances before):
//   * 17 geometries: 8x16, 8x20, 16x16 and 32x16 never win the cost model (Engine::choose_plan) -- a geometry of the
//     same row capacity with ~10 rows per lane is cheaper at every shape, for throughput and for latency;
//   * every geometry carries the 14 score kernels and the 5 alignment-fill kernels the engine picks for ordinary
//     scorings with the default tie-breaks (tagged cells; symmetric affine scores: VALIGN_FAST_KERNELS);
//   * everything else -- the equality-test fill kernels (the fallback for scorings whose tagged cells would leave int16,
//     and for the debug switch no_tag), the per-row arg-max forms, the SSE2 / AVX2 tie-break kernels and affine
//     alignments with different scores per direction -- exists for SIX geometries only, one per row capacity (64, 160,
//     320, 512, 1024, 2048); a call that needs one of them on another geometry is re-planned onto the next full one
//     (Engine::align_plan_for): same results, a sweep a few per cent longer.
#pragma once

#include "dp_kernels.hip.h"
#include "trace_kernels.hip.h"

// X(G, K): geometries with every kernel;  Y(G, K): geometries with the fast set only.  Parts are balanced by rows per
// lane (compile time grows with K).
#define VALIGN_PART0(X, Y) X(64, 32) Y(8, 4)
#define VALIGN_PART1(X, Y) Y(64, 24) Y(8, 6) Y(16, 4)
#define VALIGN_PART2(X, Y) X(64, 16) Y(8, 10)
#define VALIGN_PART3(X, Y) X(16, 10) Y(8, 12) Y(16, 8)
#define VALIGN_PART4(X, Y) X(32, 10) Y(16, 12) Y(64, 12)
#define VALIGN_PART5(X, Y) X(64, 8) X(8, 8) Y(32, 8) Y(32, 12)
#define VALIGN_KERNEL_PARTS 6

#define VALIGN_ALL_PARTS(X, Y) VALIGN_PART0(X, Y) VALIGN_PART1(X, Y) VALIGN_PART2(X, Y) VALIGN_PART3(X, Y) VALIGN_PART4(X, Y) VALIGN_PART5(X, Y)

// The kernels every geometry has; PREFIX is `extern template` (declaration) or `template` (definition).
#define VALIGN_FAST_KERNELS(PREFIX, G, K)                                                            \
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
    PREFIX __global__ void align_fill_affine_tag_kernel<G, K, kAlgSW, true>(const FillArgs);         \
    PREFIX __global__ void align_fill_affine_tag_kernel<G, K, kAlgNW, true>(const FillArgs);         \
    PREFIX __global__ void align_fill_tag_kernel<G, K, kAlgSW, true, false>(const FillArgs);         \
    PREFIX __global__ void align_fill_tag_kernel<G, K, kAlgSW, true, false, false, true>(const FillArgs); \
    PREFIX __global__ void align_fill_tag_kernel<G, K, kAlgNW, false, false>(const FillArgs);

// ... and what only the full geometries add: equality-test pointer kernels and the per-row arg-max forms of the tagged ones
#define VALIGN_FALLBACK_KERNELS(PREFIX, G, K)                                                        \
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
    PREFIX __global__ void align_fill_tag_kernel<G, K, kAlgSW, false, false>(const FillArgs);        \
    PREFIX __global__ void align_fill_tag_kernel<G, K, kAlgSW, false, true>(const FillArgs);         \
    PREFIX __global__ void align_fill_tag_kernel<G, K, kAlgSW, true, true>(const FillArgs);          \
    PREFIX __global__ void align_fill_tag_kernel<G, K, kAlgNW, false, true>(const FillArgs);         \
    PREFIX __global__ void align_fill_affine_tag_kernel<G, K, kAlgSW, false>(const FillArgs);        \
    PREFIX __global__ void align_fill_affine_tag_kernel<G, K, kAlgNW, false>(const FillArgs);
