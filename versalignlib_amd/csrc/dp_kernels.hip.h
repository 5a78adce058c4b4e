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
//     travels to lane l+1 -- one DPP wave_shr:1 per step, no LDS traffic for DP state.
//   * Rows are padded at the TOP (rows before the read starts carry class "none",
//     substitution 0): with non-positive gap scores those rows stay exactly at the
//     row-0 boundary value, so the real last row is always the last register of the
//     last lane and no per-cell row masking is needed.
//   * Substitution scores come from a per-pair query profile in LDS (5 classes x
//     padded rows, int16), laid out so that a lane fetches the scores of its K rows
//     for the current reference base with ds_read_b64 / ds_read_b32; reference bases
//     are staged once as class codes in LDS.  Inputs are raw ASCII as delivered by
//     the ABI (1 byte per base), fetched from HBM with 16-byte coalesced loads.
//   * Lanes outside [0, F) columns are EXEC-masked, so finished lanes keep the values
//     of the last column (needed by the NW-variant result).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace valign {

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

constexpr int kWave = 64;
constexpr int kAlgSW = 0;
constexpr int kAlgNW = 1;
constexpr short kNegInf = -16384;   // "minus infinity" of the affine NW borders (oracle: NEG_INF)

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

// Geometry of one kernel instantiation.
template <int G, int K>
struct Geo {
    static_assert(G == 4 || G == 8 || G == 16 || G == 32 || G == 64, "group size");
    static_assert(K % 2 == 0, "rows per lane must be even");
    static constexpr int kGroups = kWave / G;        // lane groups per wave
    static constexpr int kPairs = 2 * kGroups;       // pairs per wave
    static constexpr int kRows = G * K;              // padded rows
    static constexpr int kChunks = K / 4;            // 4-row chunks read as b64
    static constexpr int kRem = K % 4;               // 0 or 2 rows read as b32
    static constexpr int kPairStride = kRows * 2;    // bytes of one (class, pair) array
    static constexpr int kClassStride = kPairs * kPairStride;
    static constexpr int kProfBytes = 5 * kClassStride;
    // byte offset of row q of lane l inside one (class, pair) array
    __host__ __device__ static constexpr int row_offset(int l, int q) {
        return q < 4 * kChunks ? (q / 4) * (G * 8) + l * 8 + (q % 4) * 2
                               : kChunks * (G * 8) + l * 4 + (q - 4 * kChunks) * 2;
    }
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
    unsigned char *prof;    // [5 classes][pairs][lane rows] int16 substitution scores
    unsigned char *refc;    // [groups][2 * F] class codes, pair A / pair B interleaved
    int *first_bad;         // [pairs][2]: first read / ref position whose class is 0 (else R / F)
    long long pair0;        // first pair of this wave
    int last;               // index of the last existing pair of the wave (tail waves are short)
};

// Stage the wave's raw reads/refs with coalesced 16-byte loads, then build the class-code
// arrays and the query profile.  Returns false for a wave past the end of the batch
// (it still takes part in the block barriers).
template <int G, int K, bool FIND_BAD>
__device__ __forceinline__ bool wave_setup(const uint8_t *reads, const uint8_t *refs, long long n, int R, int F,
                                           int prof_area, int refc_stride, int wave_lds, short match,
                                           short mismatch, WaveTables &w) {
    using geo = Geo<G, K>;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int pad_rows = geo::kRows - R;

    unsigned char *lds = valign_smem + (size_t)wave * wave_lds;
    unsigned char *prof = lds;
    unsigned char *refc = lds + prof_area;
    unsigned char *rstage = refc + geo::kGroups * refc_stride;   // raw reads of the wave
    int *first_bad = reinterpret_cast<int *>(lds + wave_lds - geo::kPairs * 8);

    const int waves = blockDim.x / kWave;
    const long long pair0 = ((long long)blockIdx.x * waves + wave) * geo::kPairs;
    const bool live = pair0 < n;
    long long pair_end = pair0 + geo::kPairs;
    if (pair_end > n) pair_end = n;
    const int last = (int)(pair_end - pair0) - 1;

    // ---- stage raw bytes: refs into the (not yet built) profile area, reads beside ----
    long long ref_lo = 0, read_lo = 0;
    if (live) {
        ref_lo = pair0 * F;
        read_lo = pair0 * R;
        stage_span(prof, refs, ref_lo, pair_end * F, lane);
        stage_span(rstage, reads, read_lo, pair_end * R, lane);
        if (FIND_BAD && lane < geo::kPairs) {
            first_bad[2 * lane] = R;
            first_bad[2 * lane + 1] = F;
        }
    }
    __syncthreads();
    const int ref_skew = (int)((unsigned long long)(refs + ref_lo) & 15ull);
    const int read_skew = (int)((unsigned long long)(reads + read_lo) & 15ull);

    // ---- reference bases -> profile class (0..3 = A,T,C,G; 4 = scores nothing) ----
    if (live) {
        for (int idx = lane; idx < geo::kGroups * F; idx += kWave) {
            const int g = idx / F, j = idx - g * F;
            int pa = 2 * g, pb = 2 * g + 1;
            pa = pa > last ? last : pa;
            pb = pb > last ? last : pb;
            const int ca = base_class(prof[ref_skew + pa * F + j]);
            const int cb = base_class(prof[ref_skew + pb * F + j]);
            unsigned char *dst = refc + g * refc_stride + 2 * j;
            dst[0] = (unsigned char)((ca >= 1 && ca <= 4) ? ca - 1 : 4);
            dst[1] = (unsigned char)((cb >= 1 && cb <= 4) ? cb - 1 : 4);
            if (FIND_BAD) {
                if (ca == 0) atomicMin(&first_bad[2 * (2 * g) + 1], j);
                if (cb == 0) atomicMin(&first_bad[2 * (2 * g + 1) + 1], j);
            }
        }
    }
    __syncthreads();

    // ---- query profile: prof[class][pair][lane rows] = S(read base of the row, class) ----
    if (live) {
        for (int idx = lane; idx < geo::kPairs * geo::kRows; idx += kWave) {
            const int p = idx / geo::kRows, rr = idx - p * geo::kRows;
            const int ps = p > last ? last : p;
            int a = 0;
            if (rr >= pad_rows) {
                a = base_class(rstage[read_skew + ps * R + (rr - pad_rows)]);
                if (FIND_BAD && a == 0) atomicMin(&first_bad[2 * p], rr - pad_rows);
            }
            const bool valid = a >= 1 && a <= 4;
            const int off = p * geo::kPairStride + geo::row_offset(rr / K, rr % K);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const short sc = valid ? (a == c + 1 ? match : mismatch) : (short)0;
                *reinterpret_cast<short *>(prof + c * geo::kClassStride + off) = sc;
            }
            *reinterpret_cast<short *>(prof + 4 * geo::kClassStride + off) = 0;
        }
    }
    __syncthreads();
    w.prof = prof;
    w.refc = refc;
    w.first_bad = first_bad;
    w.pair0 = pair0;
    w.last = last;
    return live;
}

// S[q] (q = 0..K-1): substitution scores of this lane's K rows against the current reference
// bases of pair A (low halves) and pair B (high halves), from the LDS profile.
template <int G, int K>
__device__ __forceinline__ void fetch_profile(unsigned addr_a, unsigned addr_b, int rem_delta, s16x2 (&S)[K]) {
    using geo = Geo<G, K>;
#pragma unroll
    for (int c = 0; c < geo::kChunks; ++c) {
        const uint2 va = *reinterpret_cast<const uint2 *>(valign_smem + addr_a + c * (G * 8));
        const uint2 vb = *reinterpret_cast<const uint2 *>(valign_smem + addr_b + c * (G * 8));
        S[4 * c + 0] = as_pk(__builtin_amdgcn_perm(vb.x, va.x, 0x05040100u));
        S[4 * c + 1] = as_pk(__builtin_amdgcn_perm(vb.x, va.x, 0x07060302u));
        S[4 * c + 2] = as_pk(__builtin_amdgcn_perm(vb.y, va.y, 0x05040100u));
        S[4 * c + 3] = as_pk(__builtin_amdgcn_perm(vb.y, va.y, 0x07060302u));
    }
    if (geo::kRem) {
        const unsigned va = *reinterpret_cast<const unsigned *>(valign_smem + addr_a + rem_delta);
        const unsigned vb = *reinterpret_cast<const unsigned *>(valign_smem + addr_b + rem_delta);
        S[K - 2] = as_pk(__builtin_amdgcn_perm(vb, va, 0x05040100u));
        S[K - 1] = as_pk(__builtin_amdgcn_perm(vb, va, 0x07060302u));
    }
}

template <int G, int K, int ALG, bool AFFINE>
__global__ void __launch_bounds__(256)
score_kernel(const ScoreArgs args) {
    using geo = Geo<G, K>;
    const int lane = threadIdx.x & (kWave - 1);
    const int grp = lane / G;
    const int l = lane % G;
    const int F = args.F;

    WaveTables w;
    if (!wave_setup<G, K, false>(args.reads, args.refs, args.n, args.R, F, args.prof_area, args.refc_stride,
                                 args.wave_lds, args.match, args.mismatch, w))
        return;
    unsigned char *prof = w.prof;
    unsigned char *refc = w.refc;
    const long long pair0 = w.pair0;

    // ---- per-lane constants ----
    const unsigned lmask = l == 0 ? 0u : 0xFFFFFFFFu;            // group leader: row-0 border
    const unsigned prof_base = (unsigned)(prof - valign_smem);
    const unsigned lane_a = prof_base + (2 * grp) * geo::kPairStride + l * 8;
    const unsigned lane_b = lane_a + geo::kPairStride;
    const int rem_delta = geo::kChunks * (G * 8) - l * 4;        // b32 tail chunk: lane stride 4
    const unsigned char *codes = refc + grp * args.refc_stride - 2 * l;

    s16x2 g_read, g_ref, o_read, e_read, o_ref, e_ref;
    if (ALG == kAlgSW) {           // magnitudes for the unsigned floor-at-zero subtract
        g_read = pk((short)-args.gap_read);   g_ref = pk((short)-args.gap_ref);
        o_read = pk((short)-args.open_read);  e_read = pk((short)-args.ext_read);
        o_ref = pk((short)-args.open_ref);    e_ref = pk((short)-args.ext_ref);
    } else {                       // signed addends
        g_read = pk(args.gap_read);   g_ref = pk(args.gap_ref);
        o_read = pk(args.open_read);  e_read = pk(args.ext_read);
        o_ref = pk(args.open_ref);    e_ref = pk(args.ext_ref);
    }
    const s16x2 border_f = pk(ALG == kAlgNW ? kNegInf : (short)0);

    s16x2 Hl[K], El[K];
#pragma unroll
    for (int q = 0; q < K; ++q) {
        Hl[q] = pk(0);
        El[q] = border_f;
    }
    s16x2 up0 = pk(0), h_last = pk(0), f_last = border_f, best = pk(0), row_best = pk(0);

    const int steps = F + G - 1;
    for (int t = 0; t < steps; ++t) {
        const s16x2 diag0 = up0;
        up0 = as_pk(from_prev_lane(as_u32(h_last)) & lmask);
        s16x2 fup0 = border_f;
        if (AFFINE) {
            const unsigned fv = from_prev_lane(as_u32(f_last));
            fup0 = (ALG == kAlgNW) ? as_pk(l == 0 ? as_u32(border_f) : fv) : as_pk(fv & lmask);
        }
        const int j = t - l;
        if ((unsigned)j < (unsigned)F) {
            const unsigned ca = codes[2 * t], cb = codes[2 * t + 1];
            const unsigned addr_a = lane_a + ca * geo::kClassStride;
            const unsigned addr_b = lane_b + cb * geo::kClassStride;
            s16x2 S[K];
            fetch_profile<G, K>(addr_a, addr_b, rem_delta, S);
            // pass 1: everything that only needs the previous column
            s16x2 m[K];
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const s16x2 d = (q == 0 ? diag0 : Hl[q - 1]) + S[q];
                s16x2 e;
                if (AFFINE) {
                    e = (ALG == kAlgSW)
                            ? pk_max(pk_sub_floor0(El[q], e_read), pk_sub_floor0(Hl[q], o_read))
                            : pk_max(pk_add_sat(El[q], e_read), pk_add_sat(Hl[q], o_read));
                    El[q] = e;
                } else {
                    e = (ALG == kAlgSW) ? pk_sub_floor0(Hl[q], g_read) : Hl[q] + g_read;
                }
                m[q] = pk_max(d, e);
            }
            // pass 2: the in-lane chain down the column
            s16x2 h = up0, f = fup0;
#pragma unroll
            for (int q = 0; q < K; ++q) {
                if (AFFINE) {
                    f = (ALG == kAlgSW) ? pk_max(pk_sub_floor0(f, e_ref), pk_sub_floor0(h, o_ref))
                                        : pk_max(pk_add_sat(f, e_ref), pk_add_sat(h, o_ref));
                } else {
                    f = (ALG == kAlgSW) ? pk_sub_floor0(h, g_ref) : h + g_ref;
                }
                h = pk_max(m[q], f);
                Hl[q] = h;
                if (ALG == kAlgSW) best = pk_max(best, h);
            }
            h_last = h;
            f_last = f;
            if (ALG == kAlgNW) row_best = pk_max(row_best, h);
        }
    }

    // ---- result ----
    s16x2 res;
    if (ALG == kAlgSW) {
        res = best;
    } else {
        // last column: every lane froze at column F-1; last row: last register of lane G-1
        s16x2 col = Hl[0];
#pragma unroll
        for (int q = 1; q < K; ++q) col = pk_max(col, Hl[q]);
        res = pk_max(col, l == G - 1 ? row_best : pk(0));
        res = pk_max(res, pk(0));
    }
#pragma unroll
    for (int d = G / 2; d >= 1; d >>= 1)
        res = pk_max(res, as_pk((unsigned)__shfl_xor((int)as_u32(res), d, kWave)));
    if (l == 0) {
        const long long pa = pair0 + 2 * grp;
        if (pa + 1 < args.n && ((unsigned long long)args.scores & 3ull) == 0) {
            *reinterpret_cast<unsigned *>(args.scores + pa) = as_u32(res);
        } else {
            if (pa < args.n) args.scores[pa] = res.x;
            if (pa + 1 < args.n) args.scores[pa + 1] = res.y;
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
    const int raw_refs = ((geo::kPairs * F + 16 + 15) / 16) * 16;    // staged raw, then overwritten
    w.prof_area = geo::kProfBytes > raw_refs ? geo::kProfBytes : raw_refs;
    w.prof_area = ((w.prof_area + 15) / 16) * 16;
    w.refc_stride = ((2 * F + 15) / 16) * 16;
    const int raw_reads = ((geo::kPairs * R + 16 + 15) / 16) * 16;
    w.total = w.prof_area + geo::kGroups * w.refc_stride + raw_reads + ((geo::kPairs * 8 + 15) / 16) * 16;
    return w;
}

}  // namespace valign
