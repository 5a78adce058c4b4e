// pack_kernels.hip.h -- the device end of the 4-bit class transfer (host_pipeline.h: pack_classes).
//
// score_alignments only looks at the class of a base (reference: char_to_score, src/Kernels/default/DefaultKernel.h
// :43-60), so the host-pointer path ships two classes per byte across PCIe and this kernel expands them in HBM to one
// canonical byte per base -- 0 -> NUL, 1 -> 'A', 2 -> 'T', 3 -> 'C', 4 -> 'G', 5 -> 'N' -- in the pair-major layout
// every score kernel reads.  The kernels then see bytes of exactly the classes the caller's bytes had: identical
// scores.  HBM-bound and tiny: 0.5 byte in, 1 byte out per base (1 M pairs of 150 x 500: 0.33 GB in, 0.65 GB out).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace valign {

struct UnpackArgs {
    const uint8_t *packed;      // n * ((len + 1) / 2) bytes, low nibble first
    uint8_t *out;               // n * len bytes
    long long n;
    int len;
};

// four nibbles (16 bits, first base lowest) -> four canonical bytes: the selector bytes of one v_perm_b32 over the
// eight-entry table {NUL, A, T, C | G, N, NUL, NUL}
__device__ __forceinline__ unsigned expand_classes4(unsigned x) {
    unsigned sel = (x | (x << 8)) & 0x00FF00FFu;
    sel = (sel | (sel << 4)) & 0x0F0F0F0Fu;
    return __builtin_amdgcn_perm(0x00004E47u, 0x43544100u, sel & 0x07070707u);
}

#ifdef VALIGN_TU_SCORE      // not templates: defined once, in engine_score.hip
// Even `len`: the batch is one flat nibble stream (row boundaries fall on byte boundaries) -- a thread expands 8
// packed bytes into 16 output bytes, 8-byte coalesced loads, 16-byte coalesced stores.
__global__ void __launch_bounds__(256)
unpack_even_kernel(const UnpackArgs a) {
    const long long total = a.n * (long long)(a.len / 2);              // packed bytes
    const long long at = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (at >= total) return;
    if (at + 8 <= total) {
        const uint2 v = *reinterpret_cast<const uint2 *>(a.packed + at);
        uint4 o;
        o.x = expand_classes4(v.x & 0xFFFFu);
        o.y = expand_classes4(v.x >> 16);
        o.z = expand_classes4(v.y & 0xFFFFu);
        o.w = expand_classes4(v.y >> 16);
        *reinterpret_cast<uint4 *>(a.out + 2 * at) = o;
    } else {
        for (long long b = at; b < total; ++b) {
            const unsigned two = expand_classes4(a.packed[b]);
            a.out[2 * b] = (uint8_t)two;
            a.out[2 * b + 1] = (uint8_t)(two >> 8);
        }
    }
}

// Odd `len`: every row ends in a half-used byte.  One thread per packed byte.
__global__ void __launch_bounds__(256)
unpack_odd_kernel(const UnpackArgs a) {
    const int plen = (a.len + 1) / 2;
    const long long at = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (at >= a.n * (long long)plen) return;
    const long long pair = at / plen;
    const int k = (int)(at - pair * plen);
    const unsigned two = expand_classes4(a.packed[at]);
    uint8_t *dst = a.out + pair * a.len + 2 * k;
    dst[0] = (uint8_t)two;
    if (2 * k + 1 < a.len) dst[1] = (uint8_t)(two >> 8);
}
#endif

}  // namespace valign
