// dp_kernels.hip.h -- gfx950 score kernels for the versalignLib AlignmentKernel path.
//
// What is computed (reference semantics, restated in oracle/cpu_ref.c):
//   Smith-Waterman score ........ src/Kernels/default/DefaultKernel.cpp:83-138
//   Needleman-Wunsch-variant .... src/Kernels/default/DefaultKernel.cpp:140-202
//   substitution classes ........ src/Kernels/default/DefaultKernel.h:43-60, 83-97
// plus an affine-gap (Gotoh) extension that the reference does not have.
//
// How (nothing here follows the reference's row-major sweeps or its OpenCL kernel):
//   * A lane GROUP of G lanes (G = 8/16/32/64, a slice of one wave64) owns TWO pairs
//     at once: every DP value is a packed 2 x int16 register, pair A in the low half,
//     pair B in the high half, so one v_pk_* instruction updates two cells.
//   * Lane l of the group owns K consecutive DP rows (register tile); the group sweeps
//     the reference left to right in a skewed (anti-diagonal) front: at step t lane l
//     works on reference column t - l.  The "left" dependency is a register, "up" is
//     the previous register of the same lane, and only the last row of each lane
//     travels to lane l+1 -- one DPP move per step, no LDS traffic for DP state.
//   * Rows are padded at the TOP (rows before the read starts carry class "none",
//     substitution 0): with non-positive gap scores those rows stay exactly at the
//     row-0 boundary value, so the real last row is always the last register of the
//     last lane and no per-cell row masking is needed.
//   * Substitution scores come from a per-pair query profile in LDS: four slabs
//     (A, T, C, G) of padded-rows x int16 per pair plus one all-zero slab shared by every
//     pair; the reference bases are staged once as slab numbers, so a lane's fetch address
//     is lane_base + slab * stride and its K scores arrive with the widest ds_read the
//     lane stride allows.  The slab stride carries the padding that a compile-time bank
//     model (profile_conflicts) finds conflict-free.  Inputs are raw ASCII as delivered
//     by the ABI (1 byte per base): refs via 16-byte coalesced loads, reads as coalesced
//     byte loads.
//   * Recurrence variants (GAPS), packed instructions per register (= per two cells) incl. the
//     profile merge: two linear gap scores 6 + maximum tracking, one shared gap score 5 +
//     tracking, affine 10, affine with the same open/extend for both directions 9.  Where every
//     cell provably stays a small integer the same recurrences run on packed HALF FLOATS, which
//     gfx950 gives a three-operand maximum (v_pk_maximum3_f16) and a free [0, 1] clamp on the
//     add: shared-gap linear SW 4 (keeps (h, max(h + g, 0)) scaled by 2^-10), symmetric affine 8,
//     affine 9 -- results identical to the int16 forms, which remain the fallback.  The SW maximum
//     is tracked on diag+S, off the dependency chain.
//   * The NW variant (no zero floor) runs every recurrence in a TILTED FRAME, cell (p, j) kept as
//     V - g_ref * p - g_read * j: gap steps (affine: extensions) cost nothing, the diagonal pays for
//     both through the query profile -- linear 3 packed instructions per register on half floats
//     (perm, add, max3), 4 on int16; symmetric affine 6 / 7.  See score_kernel.
//   * Pipeline fill/drain steps EXEC-mask lanes outside columns [0, F) (finished lanes keep
//     the values of the last column, needed by the NW-variant result); the steady phase
//     runs unmasked.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace valign {

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

constexpr int kWave = 64;
constexpr int kWaveLanes = 64;
constexpr int kAlgSW = 0;
constexpr int kAlgNW = 1;
constexpr int kCodePad = 72;          // columns of zero-slab codes before column 0 and after column F-1
constexpr short kNegInf = -16384;   // "minus infinity" of the affine NW borders (oracle: NEG_INF)

constexpr int kMaxScoreGroups = 16;
constexpr int kOneSweep = 0x7FFFFFFF;      // wave_setup: the sweep covers the whole read (padding rows on top)

struct ScoreArgs {
    const uint8_t *reads;     // n * R bytes, pair-major
    const uint8_t *refs;      // n * F bytes, pair-major
    int16_t *scores;          // n
    long long n;
    int R, F;
    int prof_area;            // bytes reserved for the profile (>= raw-ref staging it aliases)
    int refc_stride;          // bytes of one group's interleaved class-code array
    int wave_lds;             // bytes of LDS per wave
    short match, mismatch;
    short gap_read, gap_ref;                       // linear model, all <= 0
    short open_read, ext_read, open_ref, ext_ref;  // affine extension, all <= 0
    // Length-sorted batches (ragged_kernels.hip.h; Engine::ragged_finish): one launch sweeps several packed groups
    // of pairs that share the read stride R but have their own reference stride.  Blocks
    // [groups[g-1].block_end, groups[g].block_end) belong to group g; reads / refs / scores / n / F
    // above are then ignored in favour of the group's.  n_groups == 0: one plain batch.
    int n_groups;
    struct Group {
        unsigned block_end;
        int F;
        long long n;
        long long pair_ofs;         // into scores
        long long read_ofs, ref_ofs;   // bytes into reads / refs
    } groups[kMaxScoreGroups];
};

// ---- packed int16 helpers (each is one VOP3P instruction on gfx950) ----
__device__ __forceinline__ s16x2 pk(short v) { return s16x2{v, v}; }
__device__ __forceinline__ s16x2 pk_max(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ s16x2 pk_add_sat(s16x2 a, s16x2 b) { return __builtin_elementwise_add_sat(a, b); }
// max(a - g, 0) on non-negative a with magnitude g: v_pk_sub_u16 ... clamp
__device__ __forceinline__ s16x2 pk_sub_floor0(s16x2 a, s16x2 g) {
    return (s16x2)__builtin_elementwise_sub_sat((u16x2)a, (u16x2)g);
}
__device__ __forceinline__ unsigned as_u32(s16x2 v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ s16x2 as_pk(unsigned v) { return __builtin_bit_cast(s16x2, v); }

// value of lane-1 (wave-wide shift right by one lane); lane 0 receives 0
__device__ __forceinline__ unsigned from_prev_lane(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}

// value of the previous lane of the lane's G-lane group, 0 in the group's first lane.  Groups of 16 are the DPP rows:
// row_shr:1 with bound_ctrl is exactly that in ONE instruction (else: wave-wide shift, then the mask)
template <int G>
__device__ __forceinline__ unsigned group_prev_or_zero(unsigned v, unsigned lmask) {
    if (G == 16) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xF, 0xF, true);
    return from_prev_lane(v) & lmask;
}

// base class of one input byte: A/a 1, T/t 2, C/c 3, G/g 4, N/n 5, else 0
__device__ __forceinline__ int base_class(unsigned ch) {
    const unsigned u = ch & 0xDFu;            // fold case; bytes >= 0x80 never match
    int c = 0;
    c = (u == 'A') ? 1 : c;
    c = (u == 'T') ? 2 : c;
    c = (u == 'C') ? 3 : c;
    c = (u == 'G') ? 4 : c;
    c = (u == 'N') ? 5 : c;
    return c;
}

// LDS bank conflicts of one profile fetch, modelled per MI355X_MICROARCH.md section LDS: every lane of
// a group of G reads `lane_dw` dwords at dword address slab(pair of its group) * stride + l * lane_dw.
// width = dwords per load instruction (1: ds_read_b32 / read2_b32, banks mod 32, lane groups of 32;
// 2: ds_read_b64, banks mod 64, groups of 32; 4: ds_read_b128, banks mod 64, four fixed groups of 16).
// Returns the number of extra LDS cycles when every lane selects the same class (the systematic case).
constexpr int profile_conflicts(int G, int lane_dw, int width, int stride_dw) {
    const int banks = width == 1 ? 32 : 64;
    int extra = 0;
    const int ngroups = width == 4 ? 4 : 2;
    for (int grp = 0; grp < ngroups; ++grp) {
        int hits[64] = {};
        int worst = 0;
        for (int lane = 0; lane < kWaveLanes; ++lane) {
            int member = 0;
            if (width == 4) {
                const int m = lane & 31;                       // {0-3,12-15,20-27} vs the rest, per half
                const bool first = m < 4 || (m >= 12 && m < 16) || (m >= 20 && m < 28);
                member = (lane < 32 ? 0 : 2) + (first ? 0 : 1);
            } else {
                member = lane / 32;
            }
            if (member != grp) continue;
            const int l = lane % G, pair = 2 * (lane / G);
            const int dw = pair * stride_dw + l * lane_dw;
            for (int x = 0; x < width; ++x) {
                const int b = (dw + x) % banks;
                hits[b] += 1;
                worst = hits[b] > worst ? hits[b] : worst;
            }
        }
        extra += worst - 1;
    }
    return extra;
}

// Geometry of one kernel instantiation.
template <int G, int K>
struct Geo {
    static_assert(G == 4 || G == 8 || G == 16 || G == 32 || G == 64, "group size");
    static_assert(K % 2 == 0, "rows per lane must be even");
    static constexpr int kGroups = kWave / G;        // lane groups per wave
    static constexpr int kPairs = 2 * kGroups;       // pairs per wave
    static constexpr int kRows = G * K;              // padded rows
    static constexpr int kLaneBytes = K * 2;         // one lane's K int16 scores, contiguous
    static constexpr int kLoadDwords = (K * 2) % 16 == 0 ? 4 : ((K * 2) % 8 == 0 ? 2 : 1);
    // Slab stride: the rows plus the padding (in units of one load) that minimises the systematic
    // bank conflicts between the lane groups of a wave (e.g. 16x10: 80 -> 88 dwords turns an
    // always-2-way conflict between groups 0/1 and 2/3 into none).
    static constexpr int slab_dwords() {
        const int rows_dw = kRows / 2;
        int best = rows_dw, best_cost = profile_conflicts(G, K / 2, kLoadDwords, rows_dw);
        for (int pad = kLoadDwords; pad < 64 && best_cost > 0; pad += kLoadDwords) {
            const int c = profile_conflicts(G, K / 2, kLoadDwords, rows_dw + pad);
            if (c < best_cost) {
                best_cost = c;
                best = rows_dw + pad;
            }
        }
        return best;
    }
    static constexpr int kPairStride = slab_dwords() * 4;   // bytes of one (class, pair) slab
    // profile slabs: slab (class * kPairs + pair) for classes 0..3, plus ONE all-zero slab
    // shared by every pair for "this reference base scores nothing"
    static constexpr int kZeroSlab = 4 * kPairs;
    static constexpr int kProfBytes = (4 * kPairs + 1) * kPairStride;
    // byte offset of row q of lane l inside one (class, pair) array
    __host__ __device__ static constexpr int row_offset(int l, int q) { return l * kLaneBytes + q * 2; }
};

// Copy global bytes [begin, end) of `src` into LDS so that byte x lands at
// dst[x - (begin & ~15)] : 16-byte coalesced loads for the interior, bytes at the rims.
__device__ __forceinline__ void stage_span(unsigned char *dst, const uint8_t *src, long long begin,
                                           long long end, int lane) {
    const unsigned long long a0 = (unsigned long long)(src + begin) & ~15ull;
    const unsigned long long lo = (unsigned long long)(src + begin);
    const unsigned long long hi = (unsigned long long)(src + end);
    for (unsigned long long a = a0 + 16ull * lane; a < hi; a += 16ull * kWave) {
        unsigned char *d = dst + (a - a0);
        if (a >= lo && a + 16 <= hi) {
            *reinterpret_cast<uint4 *>(d) = *reinterpret_cast<const uint4 *>(a);
        } else {
            for (int b = 0; b < 16; ++b)
                if (a + b >= lo && a + b < hi) d[b] = *reinterpret_cast<const uint8_t *>(a + b);
        }
    }
}

extern __shared__ __align__(16) unsigned char valign_smem[];

// Per-wave LDS tables shared by the score and the alignment-fill kernels.
struct WaveTables {
    unsigned char *prof;    // [4 classes x pairs + 1 zero slab][lane rows] int16 substitution scores
    unsigned char *refc;    // [groups][2 * F] profile slab numbers, pair A / pair B interleaved
    int *first_bad;         // [pairs][2]: first read / ref position whose class is 0 (else R / F)
    long long pair0;        // first pair of this wave
    int last;               // index of the last existing pair of the wave (tail waves are short)
    int cols_used;          // 1 + last column where any pair of the wave has an ACGT base (wave-uniform)
};

// Stage the wave's raw refs with coalesced 16-byte loads, then build the per-column slab
// numbers and the query profile (read bases come straight from HBM, one coalesced byte
// per lane and row).  Returns false for a wave past the end of the batch (it still takes
// part in the block barriers).
template <int G, int K, bool FIND_BAD>
__device__ __forceinline__ bool wave_setup(const uint8_t *reads, const uint8_t *refs, long long n, int R, int F,
                                           int prof_area, int refc_stride, int wave_lds, short match,
                                           short mismatch, WaveTables &w, bool bad_is_non_acgt = false,
                                           unsigned block = blockIdx.x, short zero_score = 0,
                                           int strip_row0 = kOneSweep, short row_key_step = 0) {
    using geo = Geo<G, K>;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    // read position of padded row 0 of this sweep: -(padding rows) when one sweep covers the read; a row strip
    // of a longer read (strip_kernels.hip.h) starts wherever it starts, rows outside [0, R) are padding
    const int row0 = strip_row0 == kOneSweep ? R - geo::kRows : strip_row0;

    unsigned char *lds = valign_smem + (size_t)wave * wave_lds;
    unsigned char *prof = lds;
    unsigned char *refc = lds + prof_area;
    int *first_bad = reinterpret_cast<int *>(refc + geo::kGroups * refc_stride);

    const int waves = blockDim.x / kWave;
    const long long pair0 = ((long long)block * waves + wave) * geo::kPairs;
    const bool live = pair0 < n;
    long long pair_end = pair0 + geo::kPairs;
    if (pair_end > n) pair_end = n;
    const int last = (int)(pair_end - pair0) - 1;

    // ---- read bases of this lane's 2K (pair, padded row) items, straight from HBM ----
    // kPairs * kRows == 128 * K for every geometry, so each lane owns exactly 2K items.
    unsigned char rd[2 * K];
    long long ref_lo = 0;
    if (live) {
#pragma unroll
        for (int i = 0; i < 2 * K; ++i) {
            const int idx = lane + kWave * i;
            const int p = idx / geo::kRows, rr = idx - p * geo::kRows;
            const int ps = p > last ? last : p;
            const int pos = row0 + rr;
            rd[i] = (pos >= 0 && pos < R) ? reads[(pair0 + ps) * R + pos] : (unsigned char)0;
        }
        if (FIND_BAD && lane < geo::kPairs) {
            first_bad[2 * lane] = R;
            first_bad[2 * lane + 1] = F;
        }
    }

    // ---- reference bases -> profile slab of the pair (class * kPairs + pair, or the zero slab) ----
    // One lane group after the other: the group's two raw references are staged in the (not yet built) profile area -- 2 F
    // bytes instead of the wave's 2 F x groups (round 4: that staging, not the tables, was what a long reference cost first:
    // 150 x 4 000 on 16 x 10 needs 44 KB instead of 65) -- and turned into the slab numbers of its columns.
    int cols_used = 0;
    if (live) {
        // the sweep prefetches the codes of the next columns unconditionally: pad both ends
        for (int idx = lane; idx < geo::kGroups * 2 * kCodePad; idx += kWave) {
            const int g = idx / (2 * kCodePad), x = idx - g * (2 * kCodePad);
            const int col = x < kCodePad ? x : F + x;                    // [0, pad) and [F + pad, F + 2 pad)
            unsigned char *dst = refc + g * refc_stride + 2 * col;
            dst[0] = dst[1] = (unsigned char)geo::kZeroSlab;
        }
    }
    // (where all the wave's references fit the profile area anyway -- 8 x 500 bytes in 11.6 KB -- they are staged in one go:
    // one barrier instead of two per lane group, 1.5 % of the 150 x 500 sweep)
    const bool whole = geo::kPairs * F + 32 <= prof_area;
    if (whole) {
        if (live) stage_span(prof, refs, pair0 * F, pair_end * F, lane);
        __syncthreads();
    }
#pragma unroll
    for (int g = 0; g < geo::kGroups; ++g) {
        int pa = 2 * g, pb = 2 * g + 1;
        pa = pa > last ? last : pa;
        pb = pb > last ? last : pb;
        const long long ref_lo = whole ? pair0 * F : (pair0 + pa) * F;
        if (!whole) {
            if (live) stage_span(prof, refs, ref_lo, (pair0 + pb + 1) * F, lane);
            __syncthreads();
        }
        if (live) {
            const int ref_skew = (int)((unsigned long long)(refs + ref_lo) & 15ull);
            const unsigned char *raw_a = prof + ref_skew + (whole ? pa * F : 0), *raw_b = raw_a + (pb - pa) * F;
            unsigned char *codes_g = refc + g * refc_stride + 2 * kCodePad;
            for (int j = lane; j < F; j += kWave) {
                const int ca = base_class(raw_a[j]);
                const int cb = base_class(raw_b[j]);
                const bool va = ca >= 1 && ca <= 4, vb = cb >= 1 && cb <= 4;
                const unsigned sa = va ? (ca - 1) * geo::kPairs + 2 * g : geo::kZeroSlab;
                const unsigned sb = vb ? (cb - 1) * geo::kPairs + 2 * g + 1 : geo::kZeroSlab;
                *reinterpret_cast<unsigned short *>(codes_g + 2 * j) = (unsigned short)(sa | (sb << 8));
                if (va || vb) cols_used = j + 1 > cols_used ? j + 1 : cols_used;
                if (FIND_BAD) {
                    // "invalid" for the NW end cell: class 0 (Default kernel) or anything but ACGT (SSE kernel)
                    if (ca == 0 || (bad_is_non_acgt && ca == 5)) atomicMin(&first_bad[2 * (2 * g) + 1], j);
                    if (cb == 0 || (bad_is_non_acgt && cb == 5)) atomicMin(&first_bad[2 * (2 * g + 1) + 1], j);
                }
            }
        }
        if (!whole) __syncthreads();
    }
    if (whole) __syncthreads();

    // ---- query profile: slab[class * kPairs + pair][lane rows] = S(read base of the row, class) ----
    if (live) {
#pragma unroll
        for (int i = 0; i < 2 * K; ++i) {
            const int idx = lane + kWave * i;
            const int p = idx / geo::kRows, rr = idx - p * geo::kRows;
            const int pos = row0 + rr;
            const bool in_read = pos >= 0 && pos < R;
            const int a = in_read ? base_class(rd[i]) : 0;
            if (FIND_BAD && in_read && (a == 0 || (bad_is_non_acgt && a == 5)))
                atomicMin(&first_bad[2 * p], pos);
            const bool valid = a >= 1 && a <= 4;
            const int off = p * geo::kPairStride + geo::row_offset(rr / K, rr % K);
            // row_key_step: every score of row q of a lane also carries (15 - q) * step -- the Smith-Waterman end-cell key
            // of align_fill_tag_kernel<..., PROFKEY>, which then rides on every diagonal candidate for free
            const short row_key = (short)(row_key_step * (15 - rr % K));
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const short sc = (short)((valid ? (a == c + 1 ? match : mismatch) : zero_score) + row_key);
                *reinterpret_cast<short *>(prof + c * geo::kPairs * geo::kPairStride + off) = sc;
            }
        }
        for (int idx = lane; idx < geo::kPairStride / 4; idx += kWave) {          // dword idx: rows 2 idx, 2 idx + 1
            const unsigned lo = (unsigned short)(zero_score + row_key_step * (15 - (2 * idx) % K));
            const unsigned hi = (unsigned short)(zero_score + row_key_step * (15 - (2 * idx + 1) % K));
            reinterpret_cast<unsigned *>(prof + geo::kZeroSlab * geo::kPairStride)[idx] = lo | (hi << 16);
        }
    }
    __syncthreads();
#pragma unroll
    for (int d = kWave / 2; d >= 1; d >>= 1) {
        const int other = __shfl_xor(cols_used, d, kWave);
        cols_used = other > cols_used ? other : cols_used;
    }
    w.cols_used = __builtin_amdgcn_readfirstlane(cols_used);
    w.prof = prof;
    w.refc = refc + 2 * kCodePad;          // entry of column 0
    w.first_bad = first_bad;
    w.pair0 = pair0;
    w.last = last;
    return live;
}

typedef __attribute__((address_space(3))) const unsigned lds_cu32;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x2 lds_cu32x2;
typedef __attribute__((address_space(3))) const u32x4 lds_cu32x4;
typedef __attribute__((address_space(3))) const unsigned char lds_cu8;

// LDS byte offset of a pointer into the dynamic shared array
__device__ __forceinline__ unsigned lds_offset(const void *p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p;
}

// K/2 dwords starting at LDS byte offset `addr`, with the widest loads the lane stride allows
template <int K>
__device__ __forceinline__ void lds_load_lane(unsigned addr, unsigned (&v)[K / 2]) {
    constexpr int N = K / 2;
    if constexpr ((K * 2) % 16 == 0) {
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            const u32x4 x = *(lds_cu32x4 *)(addr + 16 * i);
            v[4 * i] = x.x; v[4 * i + 1] = x.y; v[4 * i + 2] = x.z; v[4 * i + 3] = x.w;
        }
    } else if constexpr ((K * 2) % 8 == 0) {
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            const u32x2 x = *(lds_cu32x2 *)(addr + 8 * i);
            v[2 * i] = x.x; v[2 * i + 1] = x.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = *(lds_cu32 *)(addr + 4 * i);
    }
}

// S[q] (q = 0..K-1) from the raw profile dwords of pair A (low halves) and pair B (high halves)
template <int K>
__device__ __forceinline__ void merge_profile(const unsigned (&va)[K / 2], const unsigned (&vb)[K / 2], s16x2 (&S)[K]) {
#pragma unroll
    for (int c = 0; c < K / 2; ++c) {
        S[2 * c + 0] = as_pk(__builtin_amdgcn_perm(vb[c], va[c], 0x05040100u));
        S[2 * c + 1] = as_pk(__builtin_amdgcn_perm(vb[c], va[c], 0x07060302u));
    }
}

// S[q] (q = 0..K-1): substitution scores of this lane's K rows against the current reference
// bases of pair A (low halves) and pair B (high halves), from the LDS profile.
template <int G, int K>
__device__ __forceinline__ void fetch_profile(unsigned addr_a, unsigned addr_b, s16x2 (&S)[K]) {
    unsigned va[K / 2], vb[K / 2];
    lds_load_lane<K>(addr_a, va);
    lds_load_lane<K>(addr_b, vb);
    merge_profile<K>(va, vb, S);
}

// (Round 2 also tried fetching the profile WITHOUT the merge instruction -- ds_read_u16_d16 / _d16_hi straight into the
// register halves, which removes the v_perm_b32 per register (hot loop 217 -> 197 VALU per step pair, int16 affine): 2K
// narrow LDS returns per step instead of ~K/2 wide ones cost more than the 10 % VALU they save -- int16 affine 13.10 ->
// 13.54 ms, half-float affine 11.47 -> 12.14 ms, linear 7.19 -> 8.66 ms: every LDS return writes a full wave of VGPRs
// whatever its width.  The code is gone; DESIGN.md section 3 keeps the numbers.)

// GAPS selects the recurrence: kGapLinear (two gap scores), kGapSym (linear with
// gap_read == gap_ref: one subtract serves both neighbours), kGapAffine (Gotoh extension).
constexpr int kGapLinear = 0;
constexpr int kGapSym = 1;
constexpr int kGapAffine = 2;
constexpr int kGapAffineSym = 3;   // affine with open_read == open_ref and ext_read == ext_ref
// The same recurrence as kGapAffineSym on packed half floats: every value is an
// integer of magnitude <= 2048, which fp16 represents exactly, and gfx950's v_pk_maximum3_f16
// takes three operands -- h = max3(diag + S, E, F), E = max3(E - ext, H - open, 0) (which floors the
// whole SW cell at zero) and the SW maximum tracking two rows at a time: 8.5 instead of 10 packed
// instructions per register (NW variant, tilted frame: 6 instead of 7, gap matrices start at a real -inf).  The
// engine picks it when shape x scoring stays inside +-2048.
constexpr int kGapAffineSymF16 = 4;
constexpr int kGapAffineF16 = 5;      // half floats with four different open / extend scores: 9.5 instead of 11
// Linear gaps with gap_read == gap_ref on half floats.  NW variant: h = max3(diag + S', left, up) in the tilted
// frame (score_kernel) -- perm, add, max3: 3 packed instructions per register.  Smith-Waterman needs the zero
// floor: there every value is scaled by 2^-10 (exact for integers below 1024), which turns the floor into the
// hardware clamp of v_pk_add_f16 -- max(h + g, 0) is ONE instruction, and a register pair (h, max(h + g, 0))
// per cell gives h = max3(diag + S, left', up'), left' / up' being the clamped registers: 4 per register.
constexpr int kGapSymF16 = 6;

constexpr int kTrackAll = 0, kTrackNone = 1, kTrackPair = 2;   // see score_kernel's step

// Half-float NW kernels: the constant their tilted frame is centred by -- half of the largest value a cell of the
// frame can take (best possible score plus what row and column add at the far corner).
__host__ __device__ inline int nw_frame_centre(int R, int F, int rows, int match, int mismatch, int tilt_row, int tilt_col) {
    const int best = match > mismatch ? match : mismatch;
    const int top = (R < F ? R : F) * (best > 0 ? best : 0);
    return (top + tilt_row * (rows + 1) + tilt_col * (F + 1)) / 2;
}

template <int G, int K, int ALG, int GAPS>
__global__ void __launch_bounds__(256)
score_kernel(const ScoreArgs args) {
    using geo = Geo<G, K>;
    constexpr bool F16 = GAPS == kGapAffineSymF16 || GAPS == kGapAffineF16;
    constexpr bool F16SYM = GAPS == kGapAffineSymF16;
    constexpr bool AFFINE = GAPS == kGapAffine || GAPS == kGapAffineSym || F16;
    constexpr bool SYM = GAPS == kGapSym;
    constexpr bool LINF16 = GAPS == kGapSymF16;
    constexpr bool AFFSYM = GAPS == kGapAffineSym;
    const int lane = threadIdx.x & (kWave - 1);
    const int grp = lane / G;
    const int l = lane % G;

    // the batch (or, in a length-sorted launch, the group this block belongs to): wave-uniform
    const uint8_t *reads = args.reads, *refs = args.refs;
    int16_t *scores = args.scores;
    long long n_pairs = args.n;
    int F_batch = args.F;
    unsigned block = blockIdx.x;
    if (args.n_groups > 0) {
        int g = 0;
        while (g + 1 < args.n_groups && block >= args.groups[g].block_end) ++g;
        block -= g ? args.groups[g - 1].block_end : 0u;
        reads += args.groups[g].read_ofs;
        refs += args.groups[g].ref_ofs;
        scores += args.groups[g].pair_ofs;
        n_pairs = args.groups[g].n;
        F_batch = args.groups[g].F;
    }

    WaveTables w;
    // the query profile holds the substitution scores in the cell format of the recurrence
    // (kGapSymF16: S - g, also for the rows / bases that score 0)
    // (kGapSymF16: NW S - g, also for the rows / bases that score 0; SW S * 2^-10)
    constexpr bool LINF16_SW = LINF16 && ALG == kAlgSW, LINF16_NW = LINF16 && ALG == kAlgNW;
    // The NW variant (no zero floor) runs in a tilted frame: cell (p, j) -- p padded row, j matrix column -- is kept as
    //   V'(p, j) = V(p, j) - g_ref * p - g_read * j          (g: the gap score, or the extension score with affine gaps)
    // In it a gap step (an extension) costs nothing -- "up + g" and "left + g" are simply the neighbours' registers,
    // E' = max(E', H' + open - ext), F' likewise -- and the diagonal step pays both, folded into the query profile
    // (S - g_ref - g_read, also for the rows / bases that score 0).  One or two packed instructions per register less in
    // every recurrence; results are taken out of the frame where they are read (last row: per step, last column: once).
    constexpr bool TILT = ALG == kAlgNW;
    constexpr bool HALF = F16 || LINF16;          // cells are half floats (bit patterns in the s16x2 containers)
    const int tilt_row = TILT ? -(int)(AFFINE ? args.ext_ref : args.gap_ref) : 0;
    const int tilt_col = TILT ? -(int)(AFFINE ? args.ext_read : args.gap_read) : 0;
    const int fold = tilt_row + tilt_col;
    // Half-float cells are exact up to 2048 in magnitude and the frame only ever adds: a constant taken off every cell
    // (borders start at -centre, results get it back) puts the sweep's value range [~0, top + tilt] around zero.  The
    // host checks the range with the same formula (Engine::half_float_exact).
    const int centre = (HALF && TILT) ? nw_frame_centre(args.R, F_batch, G * K, args.match, args.mismatch, tilt_row, tilt_col) : 0;
    // ... and results are collected as "value - half the best possible score": what is taken off a cell on its way out
    // of the frame, tilt - centre + rcentre, lies within +- half the frame's tilt and the difference within +- rcentre --
    // every constant and every intermediate exact
    const int rcentre = (HALF && TILT) ? nw_frame_centre(args.R, F_batch, G * K, args.match, args.mismatch, 0, 0) : 0;
    const float unit = LINF16_SW ? 1.0f / 1024.0f : 1.0f;
    auto cell = [](int v) __attribute__((always_inline)) {           // an integer in the cell format of this kernel, both halves
        return HALF ? pk(__builtin_bit_cast(short, (_Float16)v)) : pk((short)v);
    };
    auto cell_add = [](s16x2 a, s16x2 b) __attribute__((always_inline)) {
        return HALF ? __builtin_bit_cast(s16x2, __builtin_bit_cast(f16x2, a) + __builtin_bit_cast(f16x2, b)) : a + b;
    };
    auto cell_sub = [](s16x2 a, s16x2 b) __attribute__((always_inline)) {
        return HALF ? __builtin_bit_cast(s16x2, __builtin_bit_cast(f16x2, a) - __builtin_bit_cast(f16x2, b)) : a - b;
    };
    const short s_match = HALF ? __builtin_bit_cast(short, (_Float16)(((int)args.match + fold) * unit)) : (short)(args.match + fold);
    const short s_mismatch = HALF ? __builtin_bit_cast(short, (_Float16)(((int)args.mismatch + fold) * unit)) : (short)(args.mismatch + fold);
    const short s_zero = HALF ? __builtin_bit_cast(short, (_Float16)fold) : (short)fold;
    if (!wave_setup<G, K, false>(reads, refs, n_pairs, args.R, F_batch, args.prof_area, args.refc_stride,
                                 args.wave_lds, s_match, s_mismatch, w, false, block, s_zero))
        return;
    const long long pair0 = w.pair0;
    // Smith-Waterman: columns after the last ACGT base of every reference in the wave (the NUL
    // padding of ragged batches) score nothing and can never raise the maximum -- not swept.
    // The NW variant's result lives in the last column and row of the PADDED matrix: full sweep.
    const int F = (ALG == kAlgSW) ? w.cols_used : F_batch;

    // ---- per-lane constants (LDS byte offsets) ----
    const unsigned lmask = l == 0 ? 0u : 0xFFFFFFFFu;            // group leader: row-0 border
    const unsigned lane_base = lds_offset(w.prof) + l * geo::kLaneBytes;     // + slab * kPairStride
    unsigned code_addr = lds_offset(w.refc) + grp * args.refc_stride - 2 * l;   // + 2 per step

    s16x2 g_read, g_ref, o_read, e_read, o_ref, e_ref;
    if (ALG == kAlgSW) {           // magnitudes for the unsigned floor-at-zero subtract
        g_read = pk((short)-args.gap_read);   g_ref = pk((short)-args.gap_ref);
        o_read = pk((short)-args.open_read);  e_read = pk((short)-args.ext_read);
        o_ref = pk((short)-args.open_ref);    e_ref = pk((short)-args.ext_ref);
    } else {                       // tilted frame: gap steps / extensions are free, an opening costs open - extend
        g_read = g_ref = e_read = e_ref = pk(0);
        o_read = pk((short)(args.open_read - args.ext_read));
        o_ref = pk((short)(args.open_ref - args.ext_ref));
    }
    // gap matrices start at minus infinity in the NW variant (half floats have the real thing: 0xFC00)
    const s16x2 border_f = pk(ALG == kAlgNW ? (F16 ? (short)0xFC00 : kNegInf) : (short)0);
    // half-float recurrence: signed addends (open, extend <= 0)
    const _Float16 open_h = (_Float16)(int)(TILT ? args.open_ref - args.ext_ref : args.open_ref), ext_h = (_Float16)(int)args.ext_ref;
    const _Float16 open_rd = (_Float16)(int)(TILT ? args.open_read - args.ext_read : args.open_read), ext_rd = (_Float16)(int)args.ext_read;
    const f16x2 o_half = f16x2{open_h, open_h}, e_half = f16x2{ext_h, ext_h}, zero_half = f16x2{(_Float16)0, (_Float16)0};
    const f16x2 o_read_half = f16x2{open_rd, open_rd}, e_read_half = f16x2{ext_rd, ext_rd};

    // Hl: H of the previous column; El: E of the previous column; HOl (symmetric affine only):
    // H - open of the previous column, which feeds E of this column (and, within a column, F of
    // the next row), so the subtract is done once per cell instead of twice.
    s16x2 Hl[K], El[K], HOl[K];
#pragma unroll
    for (int q = 0; q < K; ++q) {
        Hl[q] = cell(tilt_row * (l * K + q) - centre);          // column 0: the zero border (in the NW frame: what the row adds)
        El[q] = border_f;
        // border H plus (SW: minus) open
        HOl[q] = (ALG == kAlgSW) ? pk(0) : (F16 ? cell_add(Hl[q], pk(__builtin_bit_cast(short, open_h))) : Hl[q] + o_ref);
    }
    s16x2 up0 = pk(0), h_last = Hl[K - 1], f_last = border_f, best = pk(0);
    s16x2 fup_keep = border_f;       // NW variant, 16-lane groups: F of the row above the lane's rows (see the step)
    // NW result: max(0, last row, last column) -- kept as "value - centre" (half floats: the sum itself may not be exact)
    s16x2 row_best = cell(-rcentre);
    int j = -l;                                                  // this lane's column at step t
    // NW frame: the all-zero row above padded row 0, as the group leader sees it (row -1, column 1 at step 0, one
    // column on per step), and what the last padded row adds at this lane's column (taken off before the row maximum)
    s16x2 top_row = pk(0), top_step = pk(0), row_tilt = pk(0);
    const s16x2 col_step = cell(tilt_col);
    if (TILT) {
        if (l == 0) {
            top_row = cell(-tilt_row + tilt_col - centre);
            top_step = col_step;                       // (stays zero in the other lanes: it is OR-ed into their row above)
            up0 = cell(-tilt_row - centre);            // row -1, column 0: the diagonal of the first cell
        }
        row_tilt = cell(tilt_row * (G * K - 1) + tilt_col * (1 - l) - centre + rcentre);
    }

    // LDS fetches run one step ahead of the arithmetic (every lane, every step: the code arrays are
    // padded): on entry to step t the raw profile dwords of step t and the slab numbers of step t+1
    // are already in registers, so no step waits for its own LDS round trips (+3 % at 16x10, +12 %
    // at one wave per SIMD).
    constexpr bool PIPE = true;
    unsigned pa[K / 2], pb[K / 2];
    s16x2 S0[K], S1[K];          // the scores of a step (two buffers: the steady loop takes two steps per trip)
    unsigned ca_next = 0, cb_next = 0;
    if (PIPE) {
        const unsigned ca = *(lds_cu8 *)(code_addr), cb = *(lds_cu8 *)(code_addr + 1);
        lds_load_lane<K>(lane_base + ca * geo::kPairStride, pa);
        lds_load_lane<K>(lane_base + cb * geo::kPairStride, pb);
        ca_next = *(lds_cu8 *)(code_addr + 2);
        cb_next = *(lds_cu8 *)(code_addr + 3);
    }

    // One step of the skewed sweep.  MASKED steps EXEC-mask lanes whose column is outside
    // [0, F) (pipeline fill and drain); in the steady phase every lane is inside.
    // TRACK selects how a step feeds the running SW maximum: kTrackAll = every diag+S of the step;
    // in the steady phase of the shared-gap kernel steps go in pairs -- the first adds nothing
    // (kTrackNone), the second adds every max(left, up) plus its last row (kTrackPair): the "left"s
    // are all cells of the first step, the "up"s all cells of the second but the last row.  K + 1
    // maxima per two steps instead of 2K.
    auto step = [&](auto masked_tag, auto track_tag, s16x2 (&S)[K], s16x2 (&Snext)[K]) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        constexpr int TRACK = decltype(track_tag)::value;
        const s16x2 diag0 = up0;
        if (G == 16) {             // row_shr:1 is exactly "previous lane of my 16-lane group, else 0"
            up0 = as_pk((unsigned)__builtin_amdgcn_update_dpp(0, (int)as_u32(h_last), 0x111, 0xF, 0xF, true));
        } else {
            up0 = as_pk(from_prev_lane(as_u32(h_last)) & lmask);
        }
        if (TILT) up0 = as_pk(as_u32(up0) | as_u32(top_row));         // (zero in every lane but the group leader)
        s16x2 fup0 = border_f;
        if (AFFINE) {
            if (G == 16 && ALG == kAlgSW) {      // (row_shr:1: the group's first lane reads 0, the Smith-Waterman border)
                fup0 = as_pk((unsigned)__builtin_amdgcn_update_dpp(0, (int)as_u32(f_last), 0x111, 0xF, 0xF, true));
            } else if (G != 16) {
                const unsigned fv = from_prev_lane(as_u32(f_last));
                fup0 = (ALG == kAlgNW) ? as_pk(l == 0 ? as_u32(border_f) : fv) : as_pk(fv & lmask);
            }
            if (G == 16 && ALG == kAlgNW) {      // the group's first lane keeps what the register held: the border, for good
                fup_keep = as_pk((unsigned)__builtin_amdgcn_update_dpp((int)as_u32(fup_keep), (int)as_u32(f_last), 0x111, 0xF, 0xF, false));
                fup0 = fup_keep;
            }
        }
        s16x2 gup0 = pk(0);        // kGapSymF16 (SW): max(h + g, 0) of the row above
        if (LINF16_SW) gup0 = as_pk(group_prev_or_zero<G>(as_u32(f_last), lmask));
        if (PIPE) {
            merge_profile<K>(pa, pb, S);                                     // step t's scores
            lds_load_lane<K>(lane_base + ca_next * geo::kPairStride, pa);     // step t+1's profile rows
            lds_load_lane<K>(lane_base + cb_next * geo::kPairStride, pb);
            ca_next = *(lds_cu8 *)(code_addr + 4);                            // step t+2's slab numbers
            cb_next = *(lds_cu8 *)(code_addr + 5);
        }
        if (!MASKED || (unsigned)j < (unsigned)F) {
            if (!PIPE) {
                const unsigned ca = *(lds_cu8 *)(code_addr), cb = *(lds_cu8 *)(code_addr + 1);
                fetch_profile<G, K>(lane_base + ca * geo::kPairStride, lane_base + cb * geo::kPairStride, S);
            }
            if (LINF16_SW) {
                // h = max3(diag + S, left', up') with x' = max(x + g, 0) kept beside x (El[] holds the x');
                // everything times 2^-10, so the floor is the clamp of the packed add
                auto hf = [](s16x2 v) __attribute__((always_inline)) { return __builtin_bit_cast(f16x2, v); };
                auto bits = [](f16x2 v) __attribute__((always_inline)) { return __builtin_bit_cast(s16x2, v); };
                const _Float16 gs = (_Float16)((int)args.gap_ref * (1.0f / 1024.0f));
                const f16x2 g_unit = f16x2{gs, gs}, zero2 = f16x2{(_Float16)0, (_Float16)0}, one2 = f16x2{(_Float16)1, (_Float16)1};
                f16x2 up_c = hf(gup0);
                f16x2 h = zero2;
                f16x2 d_cur = hf(diag0) + hf(S[0]), d_prev = zero2;
                f16x2 bestf = hf(best);
#pragma unroll
                for (int q = 0; q < K; ++q) {
                    f16x2 d_next = d_cur;
                    if (q + 1 < K) d_next = hf(Hl[q]) + hf(S[q + 1]);       // before Hl[q] is overwritten
                    h = __builtin_elementwise_maximum(__builtin_elementwise_maximum(d_cur, hf(El[q])), up_c);
                    Hl[q] = bits(h);
                    up_c = __builtin_elementwise_min(__builtin_elementwise_max(h + g_unit, zero2), one2);   // v_pk_add_f16 clamp
                    El[q] = bits(up_c);
                    if (q & 1) bestf = __builtin_elementwise_maximum(__builtin_elementwise_maximum(bestf, d_prev), d_cur);
                    else if (q == K - 1) bestf = __builtin_elementwise_maximum(bestf, d_cur);
                    d_prev = d_cur;
                    d_cur = d_next;
                }
                best = bits(bestf);
                h_last = bits(h);
                f_last = bits(up_c);
            } else if (LINF16) {
                // NW frame: h = max3(diag + S', left, up) -- perm, add, max3: three packed instructions per register
                auto hf = [](s16x2 v) __attribute__((always_inline)) { return __builtin_bit_cast(f16x2, v); };
                f16x2 h = hf(up0);
                f16x2 d_cur = hf(diag0) + hf(S[0]);
#pragma unroll
                for (int q = 0; q < K; ++q) {
                    f16x2 d_next = d_cur;
                    if (q + 1 < K) d_next = hf(Hl[q]) + hf(S[q + 1]);       // before Hl[q] is overwritten
                    h = __builtin_elementwise_maximum(__builtin_elementwise_maximum(d_cur, hf(Hl[q])), h);
                    Hl[q] = __builtin_bit_cast(s16x2, h);
                    d_cur = d_next;
                }
                h_last = __builtin_bit_cast(s16x2, h);
            } else if (F16) {
                // Registers hold two half floats (bit patterns in the s16x2 containers).  pass1(q) -- diag + S
                // and E of row q, which only need the previous column -- is written between the links of the
                // dependent chain down the column (F, H, H - open), one row ahead.
                auto hf = [](s16x2 v) __attribute__((always_inline)) { return __builtin_bit_cast(f16x2, v); };
                auto bits = [](f16x2 v) __attribute__((always_inline)) { return __builtin_bit_cast(s16x2, v); };
                f16x2 d_cur, e_cur, d_prev = zero_half;
                auto pass1 = [&](int q, f16x2 &d, f16x2 &e) __attribute__((always_inline)) {
                    d = hf(q == 0 ? diag0 : Hl[q - 1]) + hf(S[q]);
                    const f16x2 e_open = F16SYM ? hf(HOl[q]) : hf(Hl[q]) + o_read_half;
                    e = __builtin_elementwise_maximum(TILT ? hf(El[q]) : hf(El[q]) + e_read_half, e_open);
                    if (ALG == kAlgSW) e = __builtin_elementwise_maximum(e, zero_half);     // folds into one max3: floors the cell
                    El[q] = bits(e);
                };
                f16x2 f = hf(fup0);
                f16x2 ho = hf(up0) + o_half;
                f16x2 h = zero_half;
                f16x2 bestf = hf(best);
                pass1(0, d_cur, e_cur);
#pragma unroll
                for (int q = 0; q < K; ++q) {
                    f = __builtin_elementwise_maximum(TILT ? f : f + e_half, ho);
                    f16x2 d_next = zero_half, e_next = zero_half;
                    if (q + 1 < K) pass1(q + 1, d_next, e_next);        // before Hl[q] is overwritten
                    h = __builtin_elementwise_maximum(__builtin_elementwise_maximum(d_cur, e_cur), f);
                    Hl[q] = bits(h);
                    ho = h + o_half;
                    if (F16SYM) HOl[q] = bits(ho);
                    if (ALG == kAlgSW) {
                        if (q & 1) bestf = __builtin_elementwise_maximum(__builtin_elementwise_maximum(bestf, d_prev), d_cur);
                        else if (q == K - 1) bestf = __builtin_elementwise_maximum(bestf, d_cur);
                    }
                    d_prev = d_cur;
                    d_cur = d_next;
                    e_cur = e_next;
                }
                best = bits(bestf);
                h_last = bits(h);
                f_last = bits(f);
            } else if (SYM) {
                // h = max(diag + S, max(left, up) - g): one subtract for both gap directions.  The chain
                // max -> sub -> max down the column is strictly dependent; the next row's diag + S (which
                // reads the OLD left value just before it is overwritten) and the maximum tracking are
                // written between its links so that no two dependent packed instructions are adjacent.
                s16x2 h = up0;
                s16x2 d_cur = diag0 + S[0];
#pragma unroll
                for (int q = 0; q < K; ++q) {
                    const s16x2 x = pk_max(Hl[q], h);
                    s16x2 d_next = pk(0);
                    if (q + 1 < K) d_next = Hl[q] + S[q + 1];
                    const s16x2 y = (ALG == kAlgSW) ? pk_sub_floor0(x, g_ref) : x;          // (NW frame: the gap step is free)
                    if (ALG == kAlgSW && TRACK == kTrackAll) best = pk_max(best, d_cur);   // max = a diagonal arrival
                    if (ALG == kAlgSW && TRACK == kTrackPair) best = pk_max(best, x);
                    h = pk_max(d_cur, y);
                    Hl[q] = h;
                    d_cur = d_next;
                }
                if (ALG == kAlgSW && TRACK == kTrackPair) best = pk_max(best, h);
                h_last = h;
            } else {
                // Everything of a row that only needs the previous column ("pass 1": diag + S, E, their
                // maximum, the SW maximum tracking) is computed one row ahead and written between the
                // links of the dependent chain down the column (F and H), so that no two dependent
                // packed instructions are adjacent.  pass1(q) reads the OLD Hl[q-1] / Hl[q] / HOl[q].
                auto pass1 = [&](int q) __attribute__((always_inline)) -> s16x2 {
                    const s16x2 d = (q == 0 ? diag0 : Hl[q - 1]) + S[q];
                    s16x2 e;
                    if (AFFSYM) {
                        e = pk_max((ALG == kAlgSW) ? pk_sub_floor0(El[q], e_read) : El[q], HOl[q]);
                        El[q] = e;
                    } else if (AFFINE) {
                        e = (ALG == kAlgSW)
                                ? pk_max(pk_sub_floor0(El[q], e_read), pk_sub_floor0(Hl[q], o_read))
                                : pk_max(El[q], pk_add_sat(Hl[q], o_read));
                        El[q] = e;
                    } else {
                        e = (ALG == kAlgSW) ? pk_sub_floor0(Hl[q], g_read) : Hl[q];
                    }
                    if (ALG == kAlgSW) best = pk_max(best, d);
                    return pk_max(d, e);
                };
                s16x2 h = up0, f = fup0;
                s16x2 ho = pk(0);
                if (AFFSYM) ho = (ALG == kAlgSW) ? pk_sub_floor0(up0, o_ref) : pk_add_sat(up0, o_ref);
                s16x2 m_cur = pass1(0);
#pragma unroll
                for (int q = 0; q < K; ++q) {
                    if (AFFSYM) {
                        f = pk_max((ALG == kAlgSW) ? pk_sub_floor0(f, e_ref) : f, ho);
                    } else if (AFFINE) {
                        f = (ALG == kAlgSW) ? pk_max(pk_sub_floor0(f, e_ref), pk_sub_floor0(h, o_ref))
                                            : pk_max(f, pk_add_sat(h, o_ref));
                    } else {
                        f = (ALG == kAlgSW) ? pk_sub_floor0(h, g_ref) : h;
                    }
                    s16x2 m_next = pk(0);
                    if (q + 1 < K) m_next = pass1(q + 1);        // before Hl[q] is overwritten
                    h = pk_max(m_cur, f);
                    Hl[q] = h;
                    if (AFFSYM) {
                        ho = (ALG == kAlgSW) ? pk_sub_floor0(h, o_ref) : pk_add_sat(h, o_ref);
                        HOl[q] = ho;
                    }
                    m_cur = m_next;
                }
                h_last = h;
                f_last = f;
            }
            if (ALG == kAlgNW) {
                const s16x2 last = cell_sub(h_last, row_tilt);             // the last padded row's cell, out of the frame
                if (HALF) row_best = __builtin_bit_cast(s16x2, __builtin_elementwise_maximum(__builtin_bit_cast(f16x2, row_best),
                                                                                      __builtin_bit_cast(f16x2, last)));
                else row_best = pk_max(row_best, last);
            }
        }
        if (TILT) {
            top_row = cell_add(top_row, top_step);
            row_tilt = cell_add(row_tilt, col_step);
        }
        ++j;
        code_addr += 2;
    };

    const int steps = F + G - 1;
    const int fill_end = G - 1 < steps ? G - 1 : steps;
    const int steady_end = F > fill_end ? F : fill_end;
    int t = 0;
    using all_t = std::integral_constant<int, kTrackAll>;
    using first_t = std::integral_constant<int, (SYM && ALG == kAlgSW) ? kTrackNone : kTrackAll>;
    using second_t = std::integral_constant<int, (SYM && ALG == kAlgSW) ? kTrackPair : kTrackAll>;
    auto single = [&](auto masked_tag) __attribute__((always_inline)) { step(masked_tag, all_t{}, S0, S1); };
    for (; t < fill_end; ++t) single(std::true_type{});
    {                                          // two steps per trip: loop-carried registers swap roles
        for (; t + 1 < steady_end; t += 2) {   // instead of being copied (+4 % SW, +10 % NW linear,
            step(std::false_type{}, first_t{}, S0, S1);    // +4 % affine together with the pipelined fetch)
            step(std::false_type{}, second_t{}, S1, S0);
        }
    }
    for (; t < steady_end; ++t) single(std::false_type{});
    for (; t < steps; ++t) single(std::true_type{});

    // ---- result ----
    s16x2 res;
    if (LINF16_SW) {
        const f16x2 b = __builtin_bit_cast(f16x2, best);
        res = s16x2{(short)(int)((float)b.x * 1024.0f), (short)(int)((float)b.y * 1024.0f)};
    } else if (HALF) {
        f16x2 b = __builtin_bit_cast(f16x2, best);
        if (ALG == kAlgNW) {          // max(0, last column of every row, last row of every column), out of the tilt
            b = __builtin_bit_cast(f16x2, l == G - 1 ? row_best : cell(-rcentre));
#pragma unroll
            for (int q = 0; q < K; ++q)
                b = __builtin_elementwise_maximum(b, __builtin_bit_cast(f16x2, cell_sub(Hl[q], cell(tilt_row * (l * K + q) + tilt_col * F - centre + rcentre))));
            // b >= -rcentre (the zero of the result rule); the rest of the score comes back as an integer
            res = s16x2{(short)((int)b.x + rcentre), (short)((int)b.y + rcentre)};
        } else {
            res = s16x2{(short)(int)b.x, (short)(int)b.y};
        }
    } else if (ALG == kAlgSW) {
        res = best;
    } else {
        // last column: every lane froze at column F-1; last row: last register of lane G-1
        s16x2 col = l == G - 1 ? row_best : pk(0);
#pragma unroll
        for (int q = 0; q < K; ++q) col = pk_max(col, Hl[q] - cell(tilt_row * (l * K + q) + tilt_col * F));
        res = pk_max(col, pk(0));
    }
#pragma unroll
    for (int d = G / 2; d >= 1; d >>= 1)
        res = pk_max(res, as_pk((unsigned)__shfl_xor((int)as_u32(res), d, kWave)));
    if (l == 0) {
        const long long pa = pair0 + 2 * grp;
        if (pa + 1 < n_pairs && ((unsigned long long)scores & 3ull) == 0) {
            *reinterpret_cast<unsigned *>(scores + pa) = as_u32(res);
        } else {
            if (pa < n_pairs) scores[pa] = res.x;
            if (pa + 1 < n_pairs) scores[pa + 1] = res.y;
        }
    }
}

// ---- host-side geometry: LDS bytes one wave needs for shape (R, F) ----
struct WaveLds {
    int prof_area, refc_stride, total;
};

template <int G, int K>
inline WaveLds wave_lds(int R, int F) {
    using geo = Geo<G, K>;
    WaveLds w;
    const int raw_refs = ((2 * F + 16 + 15) / 16) * 16;              // one lane group's two references, staged raw, then overwritten
    w.prof_area = geo::kProfBytes > raw_refs ? geo::kProfBytes : raw_refs;
    w.prof_area = ((w.prof_area + 15) / 16) * 16;
    w.refc_stride = ((2 * (F + 2 * kCodePad) + 15) / 16) * 16;
    w.total = w.prof_area + geo::kGroups * w.refc_stride + ((geo::kPairs * 8 + 15) / 16) * 16;
    return w;
}

}  // namespace valign
