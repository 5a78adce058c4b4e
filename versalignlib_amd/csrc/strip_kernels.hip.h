// strip_kernels.hip.h -- compute_alignments for reads that do not fit one register sweep (read_length > 2048).
//
// The reference computes alignments for every shape its 16-bit coordinates allow
// (src/Kernels/default/DefaultKernel.cpp:391-456, pointer matrix :394-395).  Here the read is cut into ROW STRIPS
// of 64 * K rows; each strip is one launch of align_strip_kernel -- the anti-diagonal register sweep of
// align_fill_kernel (trace_kernels.hip.h) with a whole wave per pair-of-pairs -- and strips run one after the other
// in stream order:
//   * the bottom row of a strip (one packed dword per column: pair A low, pair B high) goes to an HBM boundary
//     row and comes back as the row above the next strip.  Lane 0 needs one value per step: the wave loads 64
//     columns at a time, coalesced, one per lane, 64 steps ahead, and each step broadcasts its column with
//     v_readlane; the last lane's values are collected one lane per column and stored 64 at a time;
//   * every strip streams its 2-bit pointers to its own region of the pointer scratch (layout of
//     trace_kernels.hip.h, one region per strip), and traceback_kernel crosses from region to region;
//   * what needs the whole pair -- the first invalid read / ref position of the NW variant's end-cell rule
//     (DefaultKernel.cpp:307-315, 348-350) -- is computed up front by first_invalid_kernel; the Smith-Waterman end
//     cell (row-major first maximum, :252-256) is the best of the strips' own, earlier strips winning ties.
// Default tie-breaks, pointers by equality tests (the tagged-cell kernels keep 4 x / 8 x cell in int16, which long reads
// outgrow): the linear model as align_fill_kernel, the affine one (AFFINE) as align_fill_affine_kernel -- E in
// registers along the row, F down the column and from strip to strip through a second boundary row, two 2-bit
// code streams (source of H; E / F extended) per cell.
#pragma once

#include "trace_kernels.hip.h"

namespace valign {

struct StripArgs {
    const uint8_t *reads;
    const uint8_t *refs;
    unsigned *ptr;              // pointer stream region of THIS strip
    EndCell *ends;              // n; read-modify-write across strips (SW), written by the owning strip (NW)
    const int *first_bad;       // n x 2: first invalid read / ref position (else R / F), first_invalid_kernel
    const unsigned *top;        // boundary row above this strip: [pair-of-pairs][row_dwords] (unused for strip 0)
    unsigned *bottom;           // boundary row below
    const unsigned *top_f;      // affine: F of the same rows
    unsigned *bottom_f;
    long long n;
    int R, F;
    int prof_area, refc_stride, wave_lds;
    int blocks8;                // 8-step blocks per lane
    int strip, strips;
    int row_dwords;             // dwords per boundary row: a multiple of 64, >= F + 135 (whole 64-column stores / loads)
    short match, mismatch;
    short gap_read, gap_ref;
    short open_read, ext_read, open_ref, ext_ref;     // affine
};

#ifdef VALIGN_TU_ALIGN      // not a template: defined once, in engine_align.hip
// First position of each read / ref whose base class is 0 (else R / F): one wave per pair.
__global__ void __launch_bounds__(64)
first_invalid_kernel(const uint8_t *reads, const uint8_t *refs, long long n, int R, int F, int *out, int n_is_invalid) {
    const long long pair = blockIdx.x;
    if (pair >= n) return;
    const int lane = threadIdx.x;
    int ir = R, jr = F;
    // "invalid": class 0 (Default kernel, DefaultKernel.cpp:308,348) -- or anything but ACGT (SSE kernel, SSEKernel.cpp:532-536)
    for (int i = lane; i < R; i += kWave) {
        const int c = base_class(reads[pair * R + i]);
        if (c == 0 || (n_is_invalid && c == 5)) {
            ir = i;
            break;
        }
    }
    for (int j = lane; j < F; j += kWave) {
        const int c = base_class(refs[pair * F + j]);
        if (c == 0 || (n_is_invalid && c == 5)) {
            jr = j;
            break;
        }
    }
#pragma unroll
    for (int d = kWave / 2; d >= 1; d >>= 1) {
        const int oi = __shfl_xor(ir, d, kWave), oj = __shfl_xor(jr, d, kWave);
        ir = oi < ir ? oi : ir;
        jr = oj < jr ? oj : jr;
    }
    if (lane == 0) {
        out[2 * pair] = ir;
        out[2 * pair + 1] = jr;
    }
}
#endif

// ---- LDS of a strip sweep (round 4): the profile of the strip's 64 K rows and a RING of reference slab numbers ----
// Up to round 4 a wave staged both references whole (2 F bytes) and kept their slab numbers for every column (2 F more):
// 40 KB at 10 kbp whatever K -- four waves per CU.  A lane needs the slab numbers of ONE column per step and the wave's
// lanes span 64 columns: a ring of 128 columns (256 bytes) is refilled 64 columns at a time, one lane per column, from
// bytes requested one block of 64 steps earlier.  LDS is then the profile alone -- 9.2 / 13.8 / 18.4 KB at K = 8 / 12 /
// 16 -- and what runs side by side is bounded by registers and by the pointer scratch.
constexpr int kStripRingCols = 128;
template <int K>
struct StripLds {
    using geo = Geo<64, K>;
    static constexpr int kRing = (geo::kProfBytes + 255) / 256 * 256;         // 256-aligned: the address is base | offset
    static constexpr int kTotal = kRing + 2 * kStripRingCols;
};

// Query profile of the strip's rows (as wave_setup builds it) and an all-"no base" ring.  One wave per block.
template <int K>
__device__ __forceinline__ bool strip_ring_setup(const uint8_t *reads, long long n, int R, int F, short match, short mismatch,
                                                 int strip_row0, WaveTables &w) {
    using geo = Geo<64, K>;
    const int lane = threadIdx.x;
    const long long pair0 = (long long)blockIdx.x * geo::kPairs;
    if (pair0 >= n) return false;
    long long pair_end = pair0 + geo::kPairs;
    if (pair_end > n) pair_end = n;
    const int last = (int)(pair_end - pair0) - 1;
    unsigned char *prof = valign_smem;
    unsigned char *ring = valign_smem + StripLds<K>::kRing;
#pragma unroll
    for (int i = 0; i < 2 * K; ++i) {                   // kPairs * kRows = 128 K items: 2 K per lane
        const int idx = lane + kWave * i;
        const int p = idx / geo::kRows, rr = idx - p * geo::kRows;
        const int ps = p > last ? last : p;
        const int pos = strip_row0 + rr;
        const int a = (pos >= 0 && pos < R) ? base_class(reads[(pair0 + ps) * R + pos]) : 0;
        const bool valid = a >= 1 && a <= 4;
        const int off = p * geo::kPairStride + geo::row_offset(rr / K, rr % K);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            *reinterpret_cast<short *>(prof + c * geo::kPairs * geo::kPairStride + off) = valid ? (a == c + 1 ? match : mismatch) : (short)0;
    }
    for (int idx = lane; idx < geo::kPairStride / 4; idx += kWave) reinterpret_cast<unsigned *>(prof + geo::kZeroSlab * geo::kPairStride)[idx] = 0u;
    for (int idx = lane; idx < 2 * kStripRingCols / 4; idx += kWave)
        reinterpret_cast<unsigned *>(ring)[idx] = 0x01010101u * (unsigned)geo::kZeroSlab;
    __syncthreads();
    w.prof = prof;
    w.refc = ring;
    w.first_bad = nullptr;
    w.pair0 = pair0;
    w.last = last;
    w.cols_used = F;                                    // (no scan of the references: Smith-Waterman sweeps every column)
    return true;
}

// The ring's side of a sweep: `commit` files the slab numbers of columns [t, t + 64) -- lane -> column t + lane -- from the
// bytes requested a block earlier, `request` asks for the next 64 columns' bytes.
struct StripRefBytes {
    unsigned a, b;
};
template <int K>
__device__ __forceinline__ void strip_ring_commit(unsigned char *ring, int col, int F, StripRefBytes raw) {
    using geo = Geo<64, K>;
    const int ca = col < F ? base_class(raw.a) : 0, cb = col < F ? base_class(raw.b) : 0;
    const unsigned sa = (ca >= 1 && ca <= 4) ? (unsigned)(ca - 1) * geo::kPairs : (unsigned)geo::kZeroSlab;
    const unsigned sb = (cb >= 1 && cb <= 4) ? (unsigned)(cb - 1) * geo::kPairs + 1u : (unsigned)geo::kZeroSlab;
    *reinterpret_cast<unsigned short *>(ring + 2 * (col & (kStripRingCols - 1))) = (unsigned short)(sa | (sb << 8));
}
__device__ __forceinline__ StripRefBytes strip_ring_request(const uint8_t *ref_a, const uint8_t *ref_b, int col, int F) {
    StripRefBytes r;
    r.a = col < F ? ref_a[col] : 0u;
    r.b = col < F ? ref_b[col] : 0u;
    return r;
}

// SSE: the tie-breaks of the reference's SSE2 / AVX2 kernels (traceback_policy = 1; linear gaps): stored states 3 DIAG
// (only between two ACGT bases) > 2 LEFT > 1 UP > 0 START, no zero-floor arithmetic on the gap terms (a floored cell
// whose neighbours lie below zero is START), as align_fill_sse_kernel (src/Kernels/AVX-SSE/SSEKernel.cpp:366-379, 646-659).
template <int K, int ALG, bool AFFINE = false, bool SSE = false>
__global__ void __launch_bounds__(64)
align_strip_kernel(const StripArgs args) {
    static_assert(!(AFFINE && SSE), "the SSE / AVX kernels have linear gaps only");
    constexpr int W = AFFINE ? 2 * K : K;                             // pointer words per lane and block
    constexpr int G = 64;
    using geo = Geo<G, K>;
    const int lane = threadIdx.x;
    const int l = lane;
    const int R = args.R;
    const int pad_total = args.strips * geo::kRows - R;             // padding rows above row 0, all in strip 0
    const int row0 = args.strip * geo::kRows - pad_total;           // read position of this strip's first row

    WaveTables w;
    if (!strip_ring_setup<K>(args.reads, args.n, R, args.F, args.match, args.mismatch, row0, w)) return;
    const int F = args.F;

    const unsigned lane_base = lds_offset(w.prof) + l * geo::kLaneBytes;
    // slab numbers of this lane's column: ring entry (j mod 128), two bytes (pair A, pair B)
    const unsigned ring_base = lds_offset(w.refc);
    unsigned code_addr = ring_base | ((unsigned)(-2 * l) & (2u * kStripRingCols - 1u));
    const uint8_t *ref_a = args.refs + w.pair0 * args.F, *ref_b = args.refs + (w.pair0 + (w.last >= 1 ? 1 : 0)) * args.F;
    StripRefBytes ref_raw = strip_ring_request(ref_a, ref_b, lane, F);         // columns [0, 64)

    const s16x2 g_read = pk(ALG == kAlgSW ? (short)-args.gap_read : args.gap_read);
    const s16x2 g_ref = pk(ALG == kAlgSW ? (short)-args.gap_ref : args.gap_ref);
    s16x2 one = pk(1), two = pk(2), three = pk(3), four = pk(4), fifteen = pk(15);
    asm volatile("" : "+v"(one), "+v"(two), "+v"(three), "+v"(four), "+v"(fifteen));     // keep the packed forms (see align_fill_kernel)
    const s16x2 sg_read = pk(args.gap_read), sg_ref = pk(args.gap_ref);       // SSE policy: signed gap scores, plain adds
    // affine: magnitudes for the SW floor-at-zero subtract, signed saturating addends for the NW variant
    const s16x2 o_read = pk(ALG == kAlgSW ? (short)-args.open_read : args.open_read), e_read = pk(ALG == kAlgSW ? (short)-args.ext_read : args.ext_read);
    const s16x2 o_ref = pk(ALG == kAlgSW ? (short)-args.open_ref : args.open_ref), e_ref = pk(ALG == kAlgSW ? (short)-args.ext_ref : args.ext_ref);
    const s16x2 border_f = pk((AFFINE && ALG == kAlgNW) ? kNegInf : (short)0);
    auto gap_add = [](s16x2 v, s16x2 c) __attribute__((always_inline)) {
        return (ALG == kAlgSW) ? pk_sub_floor0(v, c) : pk_add_sat(v, c);
    };

    int ir[2], jr[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const long long p = w.pair0 + (half > w.last ? w.last : half);
        ir[half] = args.first_bad[2 * p];
        jr[half] = args.first_bad[2 * p + 1];
    }

    s16x2 Hl[K], code[K], acc[K];
    s16x2 El[AFFINE ? K : 1], code_g[AFFINE ? K : 1], acc_g[AFFINE ? K : 1];
    s16x2 rb[ALG == kAlgSW ? K : 1], fc[ALG == kAlgSW ? K : 1], sel[ALG == kAlgNW ? K : 1];
    s16x2 rinv[SSE ? K : 1];                           // SSE policy: 1 where the row's read base is not one of ACGT (no DIAG there)
    short nw_seed[2] = {0, 0};
    const uint8_t *read_a = args.reads + (w.pair0 + 0) * R, *read_b = args.reads + (w.pair0 + (w.last >= 1 ? 1 : 0)) * R;
#pragma unroll
    for (int q = 0; q < K; ++q) {
        const int pos = row0 + l * K + q;              // read position of the row (negative: padding)
        if (SSE) {
            const int ca = (pos >= 0 && pos < R) ? base_class(read_a[pos]) : 0, cb = (pos >= 0 && pos < R) ? base_class(read_b[pos]) : 0;
            rinv[q] = s16x2{(short)((ca >= 1 && ca <= 4) ? 0 : 1), (short)((cb >= 1 && cb <= 4) ? 0 : 1)};
        }
        short border = 0;
        if (ALG == kAlgNW)                             // column 0 of the NW variant: a gap of pos + 1 read bases
            border = pos < 0 ? (short)0 : (AFFINE ? (short)(args.open_ref + pos * args.ext_ref) : (short)((pos + 1) * args.gap_ref));
        Hl[q] = pk(border);
        code[q] = pk(0);
        acc[q] = pk(0);
        if (AFFINE) {
            El[q] = border_f;
            code_g[q] = acc_g[q] = pk(0);
        }
        if (ALG == kAlgSW) {
            rb[q] = pk(0);
            fc[q] = pk(0);
        } else {
            const bool ta = ir[0] >= 1 && pos == ir[0] - 1, tb = ir[1] >= 1 && pos == ir[1] - 1;
            sel[q] = s16x2{(short)(ta ? 1 : 0), (short)(tb ? 1 : 0)};
            if (ta) nw_seed[0] = border;
            if (tb) nw_seed[1] = border;
        }
    }
    if (ALG == kAlgNW) {
        rb[0] = s16x2{nw_seed[0], nw_seed[1]};
        fc[0] = pk((short)l);
    }
    s16x2 h_last = Hl[K - 1], f_last = border_f;
    // the row above the strip at column -1 (diagonal neighbour of lane 0's first cell): column 0's border
    s16x2 up0 = pk(0);
    if (ALG == kAlgNW && l == 0 && row0 - 1 >= 0)
        up0 = pk(AFFINE ? (short)(args.open_ref + (row0 - 1) * args.ext_ref) : (short)(row0 * args.gap_ref));
    int j = -l;

    unsigned *ptr_lane = pointer_stream_lane<G, K, W>(args.ptr, w.pair0, args.blocks8, lane);
    const long long pp = w.pair0 / 2;
    const unsigned *top = args.top + pp * args.row_dwords, *top_f = args.top_f + pp * args.row_dwords;
    unsigned *bottom = args.bottom + pp * args.row_dwords, *bottom_f = args.bottom_f + pp * args.row_dwords;
    const bool has_top = args.strip > 0, has_bottom = args.strip + 1 < args.strips;
    // 64 columns of the row above per lane-register, fetched 64 steps ahead (row_dwords covers the reads)
    const unsigned border_f_bits = as_u32(border_f);
    unsigned top_cur = 0u, top_next = has_top ? top[lane] : 0u;
    unsigned topf_cur = border_f_bits, topf_next = (AFFINE && has_top) ? top_f[lane] : border_f_bits;
    unsigned bot_acc = 0u, botf_acc = 0u;

    const int steps = (ALG == kAlgSW) ? ((F + G - 1 + 7) / 8) * 8 : args.blocks8 * 8;   // whole 8-step blocks
    for (int t = 0; t < steps; ++t) {
        if ((t & 63) == 0) {
            top_cur = top_next;
            top_next = (has_top && t + 64 + lane < args.row_dwords) ? top[t + 64 + lane] : 0u;
            if (AFFINE) {
                topf_cur = topf_next;
                topf_next = (has_top && t + 64 + lane < args.row_dwords) ? top_f[t + 64 + lane] : border_f_bits;
            }
            strip_ring_commit<K>(w.refc, t + lane, F, ref_raw);                // columns [t, t + 64): lane 0 needs column t now
            ref_raw = strip_ring_request(ref_a, ref_b, t + 64 + lane, F);
        }
        const s16x2 diag0 = up0;
        const unsigned above = (unsigned)__builtin_amdgcn_readlane((int)top_cur, t & 63);      // H(row above, column t)
        // every lane takes part in the DPP move: a lane masked off by the select would be read as 0 by its neighbour
        unsigned from_lane = from_prev_lane(as_u32(h_last));
        asm volatile("" : "+v"(from_lane));
        up0 = as_pk(l == 0 ? above : from_lane);
        s16x2 fup0 = border_f;
        if (AFFINE) {
            const unsigned above_f = (unsigned)__builtin_amdgcn_readlane((int)topf_cur, t & 63);
            unsigned f_lane = from_prev_lane(as_u32(f_last));
            asm volatile("" : "+v"(f_lane));
            fup0 = as_pk(l == 0 ? above_f : f_lane);
        }
        if (AFFINE && (unsigned)j < (unsigned)F) {
            // Gotoh recurrence with the pointers by equality tests (align_fill_affine_kernel): H code 0 DIAG / 1 from F /
            // 2 from E (priority DIAG > F > E), gap code bit 1 = E extended, bit 0 = F extended (open preferred on ties)
            const unsigned ca = *(lds_cu8 *)(code_addr), cb = *(lds_cu8 *)(code_addr + 1);
            s16x2 S[K];
            fetch_profile<G, K>(lane_base + ca * geo::kPairStride, lane_base + cb * geo::kPairStride, S);
            const s16x2 tt = pk((short)t);
            s16x2 d[K], m[K];
#pragma unroll
            for (int q = 0; q < K; ++q) {
                d[q] = (q == 0 ? diag0 : Hl[q - 1]) + S[q];
                const s16x2 e_open = gap_add(Hl[q], o_read), e_extd = gap_add(El[q], e_read);
                const s16x2 e = pk_max(e_extd, e_open);
                El[q] = e;
                code_g[q] = pk_min_u(e - e_open, one);
                m[q] = pk_max(d[q], e);
            }
            s16x2 h = up0, f = fup0, hs = pk(0);
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const s16x2 f_open = gap_add(h, o_ref), f_extd = gap_add(f, e_ref);
                f = pk_max(f_extd, f_open);
                h = pk_max(m[q], f);
                Hl[q] = h;
                const s16x2 nd = pk_min_u(h - d[q], one), nf = pk_min_u(h - f, one);
                code[q] = (s16x2)((u16x2)nd << (u16x2)nf);
                code_g[q] = pk_mad_u(code_g[q], two, pk_min_u(f - f_open, one));
                if (ALG == kAlgSW) {
                    const s16x2 changed = (rb[q] - h) >> fifteen;
                    fc[q] = as_pk((as_u32(changed) & as_u32(tt)) | (~as_u32(changed) & as_u32(fc[q])));
                    rb[q] = pk_max(rb[q], h);
                } else {
                    hs = pk_mad_u(h, sel[q], hs);
                }
            }
            if (ALG == kAlgNW) {
                const s16x2 nb = pk_max(rb[0], hs);
                const s16x2 changed = (rb[0] - nb) >> fifteen;
                fc[0] = as_pk((as_u32(changed) & as_u32(tt)) | (~as_u32(changed) & as_u32(fc[0])));
                rb[0] = nb;
            }
            h_last = h;
            f_last = f;
        }
        if (SSE && (unsigned)j < (unsigned)F) {
            const unsigned ca = *(lds_cu8 *)(code_addr), cb = *(lds_cu8 *)(code_addr + 1);
            s16x2 S[K];
            fetch_profile<G, K>(lane_base + ca * geo::kPairStride, lane_base + cb * geo::kPairStride, S);
            // 1 where the reference base of the pair is not ACGT (those columns use the zero slab)
            const s16x2 cinv = as_pk((ca == (unsigned)geo::kZeroSlab ? 1u : 0u) | (cb == (unsigned)geo::kZeroSlab ? 0x10000u : 0u));
            const s16x2 tt = pk((short)t);
            s16x2 d[K], lg[K];
#pragma unroll
            for (int q = 0; q < K; ++q) {
                d[q] = (q == 0 ? diag0 : Hl[q - 1]) + S[q];
                lg[q] = Hl[q] + sg_read;
            }
            s16x2 h = up0, hs = pk(0);
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const s16x2 ug = h + sg_ref;
                h = pk_max(pk_max(d[q], lg[q]), ug);
                if (ALG == kAlgSW) h = pk_max(h, pk(0));
                Hl[q] = h;
                const s16x2 nu = pk_min_u(h - ug, one), nl = pk_min_u(h - lg[q], one);
                const s16x2 ndv = pk_max(pk_max(pk_min_u(h - d[q], one), rinv[q]), cinv);
                const s16x2 t1 = pk_mad_u(nl, nu, nl);
                code[q] = three - pk_mad_u(ndv, t1, ndv);          // 3 DIAG, 2 LEFT, 1 UP, 0 START
                if (ALG == kAlgSW) {
                    const s16x2 changed = (rb[q] - h) >> fifteen;
                    fc[q] = as_pk((as_u32(changed) & as_u32(tt)) | (~as_u32(changed) & as_u32(fc[q])));
                    rb[q] = pk_max(rb[q], h);
                } else {
                    hs = pk_mad_u(h, sel[q], hs);
                }
            }
            if (ALG == kAlgNW) {
                const s16x2 nb = pk_max(rb[0], hs);
                const s16x2 changed = (rb[0] - nb) >> fifteen;
                fc[0] = as_pk((as_u32(changed) & as_u32(tt)) | (~as_u32(changed) & as_u32(fc[0])));
                rb[0] = nb;
            }
            h_last = h;
        }
        if (!AFFINE && !SSE && (unsigned)j < (unsigned)F) {
            const unsigned ca = *(lds_cu8 *)(code_addr), cb = *(lds_cu8 *)(code_addr + 1);
            s16x2 S[K];
            fetch_profile<G, K>(lane_base + ca * geo::kPairStride, lane_base + cb * geo::kPairStride, S);
            const s16x2 tt = pk((short)t);
            s16x2 d[K], m[K];
#pragma unroll
            for (int q = 0; q < K; ++q) {
                d[q] = (q == 0 ? diag0 : Hl[q - 1]) + S[q];
                const s16x2 e = (ALG == kAlgSW) ? pk_sub_floor0(Hl[q], g_read) : Hl[q] + g_read;
                m[q] = pk_max(d[q], e);
            }
            s16x2 h = up0, hs = pk(0);
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const s16x2 ug = (ALG == kAlgSW) ? pk_sub_floor0(h, g_ref) : h + g_ref;
                h = pk_max(m[q], ug);
                const s16x2 nu = pk_min_u(h - ug, one);
                Hl[q] = h;
                // back pointer: 0 if h == diag + S, else 1 if it came from above, else 2 (DIAG > UP > LEFT)
                const s16x2 nd = pk_min_u(h - d[q], one);
                code[q] = (s16x2)((u16x2)nd << (u16x2)nu);
                if (ALG == kAlgSW) {
                    const s16x2 changed = (rb[q] - h) >> fifteen;          // 0xFFFF where h beats the row best
                    fc[q] = as_pk((as_u32(changed) & as_u32(tt)) | (~as_u32(changed) & as_u32(fc[q])));
                    rb[q] = pk_max(rb[q], h);
                } else {
                    hs = pk_mad_u(h, sel[q], hs);                          // picks the cell of the one tracked row
                }
            }
            if (ALG == kAlgNW) {
                const s16x2 nb = pk_max(rb[0], hs);
                const s16x2 changed = (rb[0] - nb) >> fifteen;
                fc[0] = as_pk((as_u32(changed) & as_u32(tt)) | (~as_u32(changed) & as_u32(fc[0])));
                rb[0] = nb;
            }
            h_last = h;
        }
#pragma unroll
        for (int q = 0; q < K; ++q) acc[q] = pk_mad_u(acc[q], four, code[q]);
        if (AFFINE) {
#pragma unroll
            for (int q = 0; q < K; ++q) acc_g[q] = pk_mad_u(acc_g[q], four, code_g[q]);
        }
        if ((t & 7) == 7) {
            if constexpr (AFFINE) {               // K words of H codes followed by K words of gap codes
                unsigned w8[2 * K];
#pragma unroll
                for (int q = 0; q < K; ++q) w8[q] = as_u32(acc[q]);
#pragma unroll
                for (int q = 0; q < K; ++q) w8[K + q] = as_u32(acc_g[q]);
                store_block_words<2 * K>(pointer_stream_block<2 * K>(ptr_lane, t >> 3), w8);
            } else {
                finish_block<K>(ptr_lane, t >> 3, acc);
            }
        }
        // bottom row of the strip: lane 63 finished column t - 63
        if (has_bottom) {
            const int col = t - (G - 1);
            if (col >= 0) {
                const int v = __builtin_amdgcn_readlane((int)as_u32(h_last), G - 1);
                bot_acc = lane == (col & 63) ? (unsigned)v : bot_acc;
                if ((col & 63) == 63 || t == steps - 1) bottom[(col & ~63) + lane] = bot_acc;
                if (AFFINE) {
                    const int vf = __builtin_amdgcn_readlane((int)as_u32(f_last), G - 1);
                    botf_acc = lane == (col & 63) ? (unsigned)vf : botf_acc;
                    if ((col & 63) == 63 || t == steps - 1) bottom_f[(col & ~63) + lane] = botf_acc;
                }
            }
        }
        ++j;
        code_addr = ((code_addr + 2u) & (2u * kStripRingCols - 1u)) | ring_base;
    }

    // ---- end cell ----
    const int strip_pad = args.strip * geo::kRows;          // padded row of this strip's first row
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const long long pair = w.pair0 + half;
        if constexpr (ALG == kAlgSW) {
            int bv = 0, bq = 0, bcol = 0;
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const int v = half ? rb[q].y : rb[q].x;
                const int c = (half ? fc[q].y : fc[q].x) & 0xFFFF;
                if (v > bv) {
                    bv = v;
                    bq = q;
                    bcol = c;
                }
            }
            unsigned key = ((unsigned)bv << 16) | (unsigned)(0xFFFF - (l * K + bq));     // larger value, then smaller row
            unsigned kmax = key;
#pragma unroll
            for (int dd = G / 2; dd >= 1; dd >>= 1) {
                const unsigned other = (unsigned)__shfl_xor((int)kmax, dd, kWave);
                kmax = other > kmax ? other : kmax;
            }
            const int p = 0xFFFF - (int)(kmax & 0xFFFF);
            const int win_lane = p / K;
            const int col_t = __shfl(bcol, win_lane, kWave);
            EndCell out;
            out.pad = 0;
            out.score = (short)(kmax >> 16);
            out.read_pos = (short)(strip_pad + p - pad_total);
            out.ref_pos = (short)(col_t - win_lane);
            if (out.score <= 0) {
                out.read_pos = 0;
                out.ref_pos = 0;
            }
            if (l == 0 && pair < args.n) {
                // row-major first maximum over the whole matrix: a later strip only wins with a larger value
                if (args.strip == 0 || out.score > args.ends[pair].score) args.ends[pair] = out;
            }
        } else {
            // the strip that holds the last valid read row writes the end cell (strip 0 when there is none)
            const int i_end = ir[half] - 1;
            const int owner = i_end >= 0 ? (i_end + pad_total) / geo::kRows : 0;
            if (owner != args.strip) continue;
            int arg_col = 0;
            if (i_end >= 0) {
                const int src_l = (i_end + pad_total - strip_pad) / K;
                const int mine = ((half ? fc[0].y : fc[0].x) & 0xFFFF) - l;
                arg_col = __shfl(mine, src_l, kWave);
            }
            const int last_ref = jr[half] - 1;
            EndCell out;
            out.pad = 0;
            out.score = 0;
            out.read_pos = (short)i_end;
            out.ref_pos = (short)(last_ref < arg_col ? last_ref : arg_col);
            if (l == 0 && pair < args.n) args.ends[pair] = out;
        }
    }
}

// ---- int32 cells: alignments whose cells leave int16 ----
// The reference's shorts wrap where read_length * gap_ref (the NW variant's column-0 border), a long mismatching stretch
// or a long, highly scored match leaves [-32768, 32767] (DefaultKernel.cpp:282-389 computes in short); rounds 1-2 refused
// such calls, round 3 took the NW variant with linear gaps.  Same strips, same pointer stream, same traceback -- with ONE
// pair per register: the wave sweeps the strip twice, pair A then pair B.  Pass A stores its 2-bit codes in the low
// halves of the stream's words, pass B reads them back and adds its high halves (the same lane wrote them: no ordering
// question).  Boundary rows: one int32 row set per pair and per matrix -- set `half` (linear gaps) or 2 * half (H) and
// 2 * half + 1 (F; affine), one set = the distance StripArgs.top_f - StripArgs.top.  Every mode of align_strip_kernel:
// both algorithms, linear / affine gaps (the same recurrences and equality-test pointers, plain int32 arithmetic; "minus
// infinity" is -2^29, the host bounds (R + F) * |score| below 2^28), default / SSE tie-breaks.  The Smith-Waterman end
// value does not fit EndCell.score: its high half travels in EndCell.pad (TraceArgs.wide_score).
template <int K, int ALG = kAlgNW, bool AFFINE = false, bool SSE = false>
__global__ void __launch_bounds__(64)
align_strip_wide_kernel(const StripArgs args) {
    static_assert(!(AFFINE && SSE), "the SSE / AVX kernels have linear gaps only");
    constexpr int W = AFFINE ? 2 * K : K;                 // pointer words per lane and block
    constexpr int G = 64;
    constexpr int kNinf = -(1 << 29);
    constexpr int kSets = AFFINE ? 2 : 1;                 // boundary row sets per pair
    using geo = Geo<G, K>;
    const int lane = threadIdx.x;
    const int l = lane;
    const int R = args.R;
    const int pad_total = args.strips * geo::kRows - R;
    const int row0 = args.strip * geo::kRows - pad_total;
    const int strip_pad = args.strip * geo::kRows;

    WaveTables w;
    if (!strip_ring_setup<K>(args.reads, args.n, R, args.F, args.match, args.mismatch, row0, w)) return;
    const int F = args.F;
    const unsigned lane_base = lds_offset(w.prof) + l * geo::kLaneBytes;
    const unsigned ring_base = lds_offset(w.refc);
    const uint8_t *ref_a = args.refs + w.pair0 * args.F, *ref_b = args.refs + (w.pair0 + (w.last >= 1 ? 1 : 0)) * args.F;
    // Smith-Waterman: magnitudes for the floor-at-zero subtract (as align_strip_kernel); NW variant: signed addends
    const int g_read = ALG == kAlgSW && !SSE ? -args.gap_read : args.gap_read, g_ref = ALG == kAlgSW && !SSE ? -args.gap_ref : args.gap_ref;
    const int o_read = ALG == kAlgSW ? -args.open_read : args.open_read, e_read = ALG == kAlgSW ? -args.ext_read : args.ext_read;
    const int o_ref = ALG == kAlgSW ? -args.open_ref : args.open_ref, e_ref = ALG == kAlgSW ? -args.ext_ref : args.ext_ref;
    auto gap_add = [](int v, int c) __attribute__((always_inline)) {
        return (ALG == kAlgSW) ? (int)__builtin_elementwise_sub_sat((unsigned)v, (unsigned)c) : v + c;
    };
    constexpr int border_f = (AFFINE && ALG == kAlgNW) ? kNinf : 0;
    unsigned *ptr_lane = pointer_stream_lane<G, K, W>(args.ptr, w.pair0, args.blocks8, lane);
    const long long pp = w.pair0 / 2;
    const size_t set_dwords = (size_t)(args.top_f - args.top);
    const bool has_top = args.strip > 0, has_bottom = args.strip + 1 < args.strips;
    const int steps = (ALG == kAlgSW) ? ((F + G - 1 + 7) / 8) * 8 : args.blocks8 * 8;

    for (int half = 0; half < 2; ++half) {
        const long long pair = w.pair0 + half;
        const long long p_src = w.pair0 + (half > w.last ? w.last : half);
        const int ir = args.first_bad[2 * p_src], jr = args.first_bad[2 * p_src + 1];
        const unsigned *top = args.top + (size_t)(kSets * half) * set_dwords + pp * args.row_dwords;
        unsigned *bottom = args.bottom + (size_t)(kSets * half) * set_dwords + pp * args.row_dwords;
        const unsigned *top_f = top + set_dwords;             // (affine only)
        unsigned *bottom_f = bottom + set_dwords;
        // NW variant: the one row whose arg-max the end-cell rule needs is the last valid read row
        // (DefaultKernel.cpp:307-315, 381-387)
        const int tracked = (ALG == kAlgNW && ir >= 1) ? ir - 1 + pad_total - strip_pad : -1;      // row of this strip, or outside [0, 64 K)
        const int tr_lane = tracked >= 0 ? tracked / K : -1, tr_q = tracked >= 0 ? tracked % K : -1;
        const uint8_t *read_seq = args.reads + p_src * R;

        int Hl[K], El[AFFINE ? K : 1];
        unsigned code[K], acc[K], code_g[AFFINE ? K : 1], acc_g[AFFINE ? K : 1];
        int rb[ALG == kAlgSW ? K : 1], fc[ALG == kAlgSW ? K : 1];      // SW: per-row best and the step of its first occurrence
        unsigned rinv = 0u;                                             // SSE policy: bit q set where the row's read base is not ACGT
        int row_best = 0, row_col = l;                      // NW: fc semantics of align_strip_kernel: step index of the arg-max
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const int pos = row0 + l * K + q;
            if (SSE) {
                const int c = (pos >= 0 && pos < R) ? base_class(read_seq[pos]) : 0;
                rinv |= (c >= 1 && c <= 4) ? 0u : (1u << q);
            }
            int border = 0;
            if (ALG == kAlgNW)                              // column 0: a gap of pos + 1 read bases
                border = pos < 0 ? 0 : (AFFINE ? args.open_ref + pos * args.ext_ref : (pos + 1) * args.gap_ref);
            Hl[q] = border;
            code[q] = acc[q] = 0u;
            if (AFFINE) {
                El[q] = border_f;
                code_g[q] = acc_g[q] = 0u;
            }
            if (ALG == kAlgSW) {
                rb[q] = 0;
                fc[q] = 0;
            } else if (l == tr_lane && q == tr_q) {
                row_best = border;
            }
        }
        int h_last = Hl[K - 1], f_last = border_f;
        int up0 = 0;                                        // row above the strip at column -1
        if (ALG == kAlgNW && l == 0 && row0 - 1 >= 0) up0 = AFFINE ? args.open_ref + (row0 - 1) * args.ext_ref : row0 * args.gap_ref;
        int j = -l;
        unsigned code_addr = ring_base | ((unsigned)(-2 * l) & (2u * kStripRingCols - 1u));
        StripRefBytes ref_raw = strip_ring_request(ref_a, ref_b, lane, F);     // columns [0, 64) (each pass starts the ring over)
        unsigned top_cur = 0u, top_next = has_top ? top[lane] : 0u;
        unsigned topf_cur = (unsigned)border_f, topf_next = (AFFINE && has_top) ? top_f[lane] : (unsigned)border_f;
        unsigned bot_acc = 0u, botf_acc = 0u;

        for (int t = 0; t < steps; ++t) {
            if ((t & 63) == 0) {
                top_cur = top_next;
                top_next = (has_top && t + 64 + lane < args.row_dwords) ? top[t + 64 + lane] : 0u;
                if (AFFINE) {
                    topf_cur = topf_next;
                    topf_next = (has_top && t + 64 + lane < args.row_dwords) ? top_f[t + 64 + lane] : (unsigned)border_f;
                }
                strip_ring_commit<K>(w.refc, t + lane, F, ref_raw);
                ref_raw = strip_ring_request(ref_a, ref_b, t + 64 + lane, F);
            }
            const int diag0 = up0;
            const int above = __builtin_amdgcn_readlane((int)top_cur, t & 63);
            int from_lane = __builtin_amdgcn_update_dpp(0, h_last, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
            asm volatile("" : "+v"(from_lane));
            up0 = l == 0 ? above : from_lane;
            int fup0 = border_f;
            if (AFFINE) {
                const int above_f = __builtin_amdgcn_readlane((int)topf_cur, t & 63);
                int f_lane = __builtin_amdgcn_update_dpp(0, f_last, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
                asm volatile("" : "+v"(f_lane));
                fup0 = l == 0 ? above_f : f_lane;
            }
            if ((unsigned)j < (unsigned)F) {
                const unsigned ca = *(lds_cu8 *)(code_addr), cb = *(lds_cu8 *)(code_addr + 1);
                s16x2 S[K];
                fetch_profile<G, K>(lane_base + ca * geo::kPairStride, lane_base + cb * geo::kPairStride, S);
                const bool col_inv = SSE && (half ? cb : ca) == (unsigned)geo::kZeroSlab;      // reference base not in ACGT
                int h = up0, f = fup0, d_prev = diag0;
#pragma unroll
                for (int q = 0; q < K; ++q) {
                    const int d = d_prev + (int)(half ? S[q].y : S[q].x);
                    d_prev = Hl[q];
                    int m;
                    if constexpr (AFFINE) {
                        // Gotoh recurrence, pointers by equality tests (align_strip_kernel): H code 0 DIAG / 1 from F / 2 from E
                        // (DIAG > F > E); gap code bit 1 = E extended, bit 0 = F extended (open preferred on ties)
                        const int e_open = gap_add(Hl[q], o_read), e_extd = gap_add(El[q], e_read);
                        const int e = e_extd > e_open ? e_extd : e_open;
                        El[q] = e;
                        const int f_open = gap_add(h, o_ref), f_extd = gap_add(f, e_ref);
                        f = f_extd > f_open ? f_extd : f_open;
                        m = d > e ? d : e;
                        m = f > m ? f : m;
                        code[q] = m == d ? 0u : (m == f ? 1u : 2u);
                        code_g[q] = (e != e_open ? 2u : 0u) | (f != f_open ? 1u : 0u);
                    } else if constexpr (SSE) {
                        // signed gap scores, no zero floor on the gap terms; 3 DIAG (only between two ACGT bases) > 2 LEFT > 1 UP > 0 START
                        const int lg = Hl[q] + g_read, ug = h + g_ref;
                        m = lg > ug ? lg : ug;
                        m = d > m ? d : m;
                        if (ALG == kAlgSW) m = m > 0 ? m : 0;
                        const bool diag_ok = m == d && !col_inv && !((rinv >> q) & 1u);
                        code[q] = diag_ok ? 3u : (m == lg ? 2u : (m == ug ? 1u : 0u));
                    } else {
                        const int lg = gap_add(Hl[q], g_read), ug = gap_add(h, g_ref);
                        m = lg > ug ? lg : ug;
                        m = d > m ? d : m;
                        code[q] = m == d ? 0u : (m == ug ? 1u : 2u);          // DIAG > UP > LEFT
                    }
                    h = m;
                    Hl[q] = m;
                    if (ALG == kAlgSW) {
                        if (m > rb[q]) {                                      // strictly greater: the first arg-max of the row wins
                            rb[q] = m;
                            fc[q] = t;
                        }
                    } else if (l == tr_lane && q == tr_q && m > row_best) {
                        row_best = m;
                        row_col = t;
                    }
                }
                h_last = h;
                f_last = f;
            }
#pragma unroll
            for (int q = 0; q < K; ++q) acc[q] = ((acc[q] << 2) | code[q]) & 0xFFFFu;
            if (AFFINE) {
#pragma unroll
                for (int q = 0; q < K; ++q) acc_g[q] = ((acc_g[q] << 2) | code_g[q]) & 0xFFFFu;
            }
            if ((t & 7) == 7) {
                unsigned *dst = pointer_stream_block<W>(ptr_lane, t >> 3);
                unsigned w8[W];
#pragma unroll
                for (int q = 0; q < K; ++q) w8[q] = half ? (dst[q] & 0xFFFFu) | (acc[q] << 16) : acc[q];
                if constexpr (AFFINE) {               // K words of H codes followed by K words of gap codes
#pragma unroll
                    for (int q = 0; q < K; ++q) w8[K + q] = half ? (dst[K + q] & 0xFFFFu) | (acc_g[q] << 16) : acc_g[q];
                }
                store_block_words<W>(dst, w8);
            }
            if (has_bottom) {
                const int col = t - (G - 1);
                if (col >= 0) {
                    const int v = __builtin_amdgcn_readlane(h_last, G - 1);
                    bot_acc = lane == (col & 63) ? (unsigned)v : bot_acc;
                    if ((col & 63) == 63 || t == steps - 1) bottom[(col & ~63) + lane] = bot_acc;
                    if (AFFINE) {
                        const int vf = __builtin_amdgcn_readlane(f_last, G - 1);
                        botf_acc = lane == (col & 63) ? (unsigned)vf : botf_acc;
                        if ((col & 63) == 63 || t == steps - 1) bottom_f[(col & ~63) + lane] = botf_acc;
                    }
                }
            }
            ++j;
            code_addr = ((code_addr + 2u) & (2u * kStripRingCols - 1u)) | ring_base;
        }

        // ---- end cell ----
        if constexpr (ALG == kAlgSW) {
            // row-major first maximum (DefaultKernel.cpp:252-256): the largest value, then the smallest row, then the row's
            // first column; over the strips, a later one only wins with a larger value
            int bv = 0, bq = 0, bcol = 0;
#pragma unroll
            for (int q = 0; q < K; ++q) {
                if (rb[q] > bv) {
                    bv = rb[q];
                    bq = q;
                    bcol = fc[q];
                }
            }
            int vmax = bv;
#pragma unroll
            for (int dd = G / 2; dd >= 1; dd >>= 1) {
                const int other = __shfl_xor(vmax, dd, kWave);
                vmax = other > vmax ? other : vmax;
            }
            int p = bv == vmax ? l * K + bq : 0x7FFFFFFF;
#pragma unroll
            for (int dd = G / 2; dd >= 1; dd >>= 1) {
                const int other = __shfl_xor(p, dd, kWave);
                p = other < p ? other : p;
            }
            const int win_lane = p / K;
            const int col_t = __shfl(bcol, win_lane, kWave);
            EndCell out;
            out.score = (short)(vmax & 0xFFFF);
            out.pad = (short)((unsigned)vmax >> 16);
            out.read_pos = (short)(strip_pad + p - pad_total);
            out.ref_pos = (short)(col_t - win_lane);
            if (vmax <= 0) {
                out.read_pos = 0;
                out.ref_pos = 0;
            }
            if (l == 0 && pair < args.n) {
                const EndCell prev = args.ends[pair];
                const int prev_score = (int)((unsigned)(unsigned short)prev.score | ((unsigned)(unsigned short)prev.pad << 16));
                if (args.strip == 0 || vmax > prev_score) args.ends[pair] = out;
            }
        } else {
            // the strip that holds the last valid read row writes it (strip 0 when there is none)
            const int i_end = ir - 1;
            const int owner = i_end >= 0 ? (i_end + pad_total) / geo::kRows : 0;
            if (owner == args.strip) {
                int arg_col = 0;
                if (i_end >= 0) arg_col = __shfl(row_col - l, tr_lane, kWave);
                const int last_ref = jr - 1;
                EndCell out;
                out.pad = 0;
                out.score = 0;
                out.read_pos = (short)i_end;
                out.ref_pos = (short)(last_ref < arg_col ? last_ref : arg_col);
                if (l == 0 && pair < args.n) args.ends[pair] = out;
            }
        }
    }
}

}  // namespace valign
