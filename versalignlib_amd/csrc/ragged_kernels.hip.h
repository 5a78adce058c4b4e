// ragged_kernels.hip.h -- length-sorted batching ON THE DEVICE.  The reference pads every sequence
// of a batch to one length (pad(), src/util/versalignUtil.cpp:17-33) and its kernels sweep the padding like anything else;
// trailing non-ACGT bytes score 0 against everything, so a pair's scores are those of its trimmed shape (SW: the maximum;
// NW variant: the last-row / last-column maximum runs down the diagonals of the padding unchanged --
// tests/test_oracle_golden.py).  Rounds 1-2 trimmed and binned on the HOST while gathering (11-14 ms per million pairs of
// 150 x 500 for a pass over every sequence tail: more than the skipped cells saved).  Here the batch is already in HBM:
//
//   ragged_classify_kernel   trimmed length of every read and reference -> length bin of the pair, histogram of bins
//   (host: fold small bins, lay the groups out -- a table of a few dozen entries)
//   ragged_place_kernel      pos[i] = the pair's place in the packed order (its group's offset + a rank inside the group)
//   ragged_copy_kernel       every pair copied to its place, at its group's strides (a wave per pair)
//   score_kernel ...         one launch per read class, the class's reference groups in its group table (dp_kernels.hip.h)
//   ragged_unpermute_kernel  scores[i] = packed_scores[pos[i]]
//
// Two passes over the sequences at HBM speed (0.3-0.5 ms per million pairs of 150 x 500) against the cells not swept.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace valign {

constexpr int kRaggedMaxBins = 512;        // read classes x reference classes
constexpr int kRaggedMaxGroups = 256;      // groups after folding
constexpr int kRaggedClassifyPairs = 64;   // pairs per 256-thread block of the classify kernel (16 per wave)

struct RaggedGroupDev {       // one packed group: strides and offsets into the packed buffers
    int R, F;
    long long pair_ofs;
    long long read_ofs, ref_ofs;
};

struct RaggedClassifyArgs {
    const uint8_t *reads, *refs;
    long long n;
    int R, F;
    const uint8_t *read_class;        // R + 1 entries: trimmed length -> read class
    const uint16_t *ref_class;        // F + 1 entries
    int NF, NG;                       // reference classes, bins
    uint16_t *bin;                    // n
    unsigned *counts;                 // NG, zeroed
};

struct RaggedPlace {          // where a pair goes: byte offsets into the packed buffers and its group's strides
    long long read_at, ref_at;
    int R, F;
};

struct RaggedPermuteArgs {
    const uint8_t *reads, *refs;
    long long n;
    int R, F;
    const uint16_t *bin;
    const uint16_t *group_of_bin;     // NG entries
    const RaggedGroupDev *groups;     // NL entries
    int NL;
    unsigned *cursors;                // NL, zeroed: pairs placed so far per group
    uint8_t *out_reads, *out_refs;
    int *pos;                         // n: place of pair i in the packed order (the group's pair_ofs + its rank)
    RaggedPlace *place;               // n: the same as byte offsets, for the copy
};

struct RaggedUnpermuteArgs {
    const int16_t *packed;
    const int *pos;
    int16_t *scores;
    long long n;
};

#ifdef VALIGN_TU_SCORE      // not templates: defined once, in engine_score.hip

__device__ __forceinline__ bool ragged_is_acgt(unsigned ch) {
    const unsigned u = ch & 0xDFu, t = u - 'A';                       // bytes >= 0x80 keep bit 7: never a letter
    return t < 20u && ((0x80045u >> t) & 1u);                         // A, C, G, T
}

__device__ __forceinline__ unsigned ragged_load_dword(const uint8_t *p) {        // (any alignment: sequences start at odd strides)
    unsigned v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// Trimmed length = index of the last ACGT byte + 1 (0: none).  A wave looks at 256 bytes of a sequence at a time (a dword
// per lane; the last L % 4 bytes of a sequence go with the first window, a byte per lane), from the tail; its 16 pairs go
// through a window TOGETHER -- 16 loads in flight instead of one (a pair at a time the kernel was a chain of load
// latencies).  `len[k]` < 0: still looking.
template <int PAIRS>
__device__ __forceinline__ void ragged_trimmed_lengths(const uint8_t *base, long long first, long long n, int L, int lane, int (&len)[PAIRS]) {
#pragma unroll
    for (int k = 0; k < PAIRS; ++k) len[k] = -1;
    const int body = L & ~3;                                       // bytes covered by whole dwords
    if (L & 3) {                                                   // the odd tail: bytes body .. L - 1
        unsigned v[PAIRS];
#pragma unroll
        for (int k = 0; k < PAIRS; ++k) {
            const long long i = first + k < n ? first + k : n - 1;
            v[k] = lane < (L & 3) ? base[i * L + body + lane] : 0u;
        }
#pragma unroll
        for (int k = 0; k < PAIRS; ++k) {
            const unsigned long long m = __ballot(lane < (L & 3) && ragged_is_acgt(v[k]));
            if (m) len[k] = body + (63 - __builtin_clzll(m)) + 1;
        }
    }
    for (int end = body; end > 0; end -= 256) {
        bool open = false;
#pragma unroll
        for (int k = 0; k < PAIRS; ++k) open = open || len[k] < 0;
        if (!open) break;                                          // (wave-uniform)
        const int at = end - 256 + 4 * lane;                       // this lane's dword: bytes at .. at + 3
        unsigned v[PAIRS];
#pragma unroll
        for (int k = 0; k < PAIRS; ++k) {
            const long long i = first + k < n ? first + k : n - 1;            // (beyond the batch: the last pair again, ignored)
            v[k] = at >= 0 ? ragged_load_dword(base + i * L + at) : 0u;
        }
#pragma unroll
        for (int k = 0; k < PAIRS; ++k) {
            unsigned hit = 0;                                      // bit b: byte b of the dword is one of ACGT
#pragma unroll
            for (int b = 0; b < 4; ++b) hit |= ragged_is_acgt((v[k] >> (8 * b)) & 0xFFu) ? (1u << b) : 0u;
            hit = at >= 0 ? hit : 0u;
            const unsigned long long m = __ballot(hit != 0u);
            if (len[k] < 0 && m) {
                const int top = 63 - __builtin_clzll(m);                       // the last lane that saw one
                const unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)hit, top);
                len[k] = end - 256 + 4 * top + (31 - __builtin_clz(bits)) + 1;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < PAIRS; ++k) len[k] = len[k] < 0 ? 0 : len[k];
}

__global__ void __launch_bounds__(256) ragged_classify_kernel(const RaggedClassifyArgs a) {
    __shared__ unsigned hist[kRaggedMaxBins];
    for (int i = threadIdx.x; i < a.NG; i += 256) hist[i] = 0u;
    __syncthreads();
    constexpr int kPerWave = kRaggedClassifyPairs / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long first = (long long)blockIdx.x * kRaggedClassifyPairs + wave * kPerWave;
    if (first < a.n) {                                                        // (wave-uniform)
        int rl[kPerWave], fl[kPerWave];
        ragged_trimmed_lengths<kPerWave>(a.reads, first, a.n, a.R, lane, rl);
        ragged_trimmed_lengths<kPerWave>(a.refs, first, a.n, a.F, lane, fl);
        // lane k files pair k
        int my_r = 0, my_f = 0;
#pragma unroll
        for (int k = 0; k < kPerWave; ++k) {
            my_r = lane == k ? rl[k] : my_r;
            my_f = lane == k ? fl[k] : my_f;
        }
        if (lane < kPerWave && first + lane < a.n) {
            const unsigned b = (unsigned)a.read_class[my_r] * (unsigned)a.NF + (unsigned)a.ref_class[my_f];
            a.bin[first + lane] = (uint16_t)b;
            atomicAdd(&hist[b], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.NG; i += 256)
        if (hist[i]) atomicAdd(&a.counts[i], hist[i]);
}

// Places: 256 pairs per block rank themselves inside the block through LDS, one global atomic per (block, group) reserves the
// block's run of places in the group.  pos[i] = place of pair i in the packed order.
__global__ void __launch_bounds__(256) ragged_place_kernel(const RaggedPermuteArgs a) {
    __shared__ unsigned cnt[kRaggedMaxGroups], base[kRaggedMaxGroups];
    const int t = threadIdx.x;
    for (int g = t; g < a.NL; g += 256) cnt[g] = 0u;
    __syncthreads();
    const long long i = (long long)blockIdx.x * 256 + t;
    unsigned g = 0, rank = 0;
    if (i < a.n) {
        g = a.group_of_bin[a.bin[i]];
        rank = atomicAdd(&cnt[g], 1u);
    }
    __syncthreads();
    for (int k = t; k < a.NL; k += 256)
        if (cnt[k]) base[k] = atomicAdd(&a.cursors[k], cnt[k]);
    __syncthreads();
    if (i < a.n) {
        const RaggedGroupDev grp = a.groups[g];
        const long long place = (long long)(base[g] + rank);
        a.pos[i] = (int)(grp.pair_ofs + place);
        a.place[i] = RaggedPlace{grp.read_ofs + place * grp.R, grp.ref_ofs + place * grp.F, grp.R, grp.F};
    }
}

// ... and the copy: a wave per pair (its place record first: one load, wave-uniform), a dword per lane (source and
// destination at any alignment), every load of the pair in flight before the first store
constexpr int kRaggedCopyRounds = 3;          // 768 bytes per sequence without a loop
__device__ __forceinline__ void ragged_copy_row(uint8_t *dst, const uint8_t *src, int len, int lane) {
    const int body = len & ~3;
    if (body <= 256 * kRaggedCopyRounds) {
        unsigned v[kRaggedCopyRounds];
#pragma unroll
        for (int r = 0; r < kRaggedCopyRounds; ++r) {
            const int x = r * 256 + 4 * lane;
            v[r] = x < body ? ragged_load_dword(src + x) : 0u;
        }
        const unsigned tail = lane < (len & 3) ? src[body + lane] : 0u;
#pragma unroll
        for (int r = 0; r < kRaggedCopyRounds; ++r) {
            const int x = r * 256 + 4 * lane;
            if (x < body) __builtin_memcpy(dst + x, &v[r], 4);
        }
        if (lane < (len & 3)) dst[body + lane] = (uint8_t)tail;
    } else {
        for (int x = lane; x < len; x += 64) dst[x] = src[x];
    }
}

__global__ void __launch_bounds__(256) ragged_copy_kernel(const RaggedPermuteArgs a) {
    const int lane = threadIdx.x & 63;
    const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= a.n) return;                                                     // (wave-uniform)
    const RaggedPlace p = a.place[i];                                         // one record, then the rows: no chain of table lookups
    ragged_copy_row(a.out_reads + p.read_at, a.reads + i * a.R, p.R, lane);
    ragged_copy_row(a.out_refs + p.ref_at, a.refs + i * a.F, p.F, lane);
}

__global__ void __launch_bounds__(256) ragged_unpermute_kernel(const RaggedUnpermuteArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < a.n) a.scores[i] = a.packed[a.pos[i]];
}

#endif  // VALIGN_TU_SCORE

}  // namespace valign
