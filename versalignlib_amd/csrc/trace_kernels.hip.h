// trace_kernels.hip.h -- gfx950 kernels for AlignmentKernel::compute_alignments.
//
// Reference semantics (restated in oracle/cpu_ref.c):
//   SW fill + pointers + end cell ... src/Kernels/default/DefaultKernel.cpp:204-280
//   NW fill + pointers + end cell ... src/Kernels/default/DefaultKernel.cpp:282-389
//   tracebacks ....................... src/Kernels/default/DefaultKernel.cpp:391-456, 458-525
//
// Two kernels:
//   align_fill_kernel   the score kernel's skewed lane-group sweep (dp_kernels.hip.h) that also
//                       derives every cell's back pointer with packed int16 arithmetic, packs
//                       2 bits per cell and streams them to an HBM scratch matrix (8 columns x
//                       K rows x 2 pairs = K dwords per lane every 8 steps, 16-byte stores),
//                       and keeps a per-row first-arg-max so that the reference's row-major
//                       "first strictly greater cell" (SW) and its row-arg-max rule (NW) are
//                       reproduced exactly from an anti-diagonal sweep.
//   traceback_kernel    one lane per pair walks the pointers back from the end cell and writes
//                       the two right-justified gapped rows + the four coordinates.
//
// Pointer states stored: 0 DIAG, 1 UP, 2 LEFT (priority DIAG > UP > LEFT as in the reference).
// START is never stored: for SW the traceback tracks the cell value itself (DIAG subtracts the
// substitution score, UP/LEFT the gap score) and stops when it reaches 0, which is exactly the
// reference's "cell == 0 -> START" rule; for NW START is row 0 and column 0 is UP.
#pragma once

#include "dp_kernels.hip.h"

namespace valign {

struct EndCell {          // per pair, 8 bytes
    short read_pos;       // 0-based read position of the end cell (-1: empty alignment)
    short ref_pos;        // 0-based ref position (-1: column 0)
    short score;          // SW: value of the end cell; NW: unused
    short pad;            // 0 -- or the high half of an int32 end value (align_strip_wide_kernel, TraceArgs.wide_score)
};

struct FillArgs {
    const uint8_t *reads;
    const uint8_t *refs;
    unsigned *ptr;            // pointer scratch: [wave][block of steps][lane of the wave][words per block] dwords
    EndCell *ends;            // n
    long long n;
    int R, F;
    int prof_area, refc_stride, wave_lds;
    int blocks8;              // 8-step blocks per lane = ceil((F + G - 1) / 8)
    short match, mismatch;
    short gap_read, gap_ref;
    short open_read, ext_read, open_ref, ext_ref;     // affine fill only
    // fused small-batch kernel (align_fill_tag_kernel<..., FUSED>): where the finished alignments go -- no pointer
    // scratch, no end cells, no second kernel
    uint8_t *out_rows;        // n * 2 * (R+F), need not be zeroed
    short *out_idx;           // n * 4
};

__device__ __forceinline__ s16x2 pk_min_u(s16x2 a, s16x2 b) {
    return (s16x2)__builtin_elementwise_min((u16x2)a, (u16x2)b);
}
__device__ __forceinline__ s16x2 pk_mad_u(s16x2 a, s16x2 b, s16x2 c) {
    return (s16x2)((u16x2)a * (u16x2)b + (u16x2)c);
}

struct TraceArgs {
    const uint8_t *reads;
    const uint8_t *refs;
    const unsigned *ptr;
    const EndCell *ends;
    uint8_t *rows;            // n * 2 * (R+F), pre-zeroed
    short *idx;               // n * 4: readStart, readEnd, refStart, refEnd
    long long n;
    int R, F;
    int G, K, pad_rows, blocks8;
    int alg;
    int affine;               // 1: pointer blocks hold K H-code words followed by K gap-code words
    int sse_policy;           // 1: stored states are 0 START, 1 UP, 2 LEFT, 3 DIAG (SSE/AVX kernel rules)
    int tagged;               // 1: codes are the tags of align_fill_tag_kernel (2 DIAG, 1 UP, 0 LEFT);
                              // 2: 4-bit codes of align_fill_affine_tag_kernel, K words per 4-step block
    short match, mismatch, gap_read, gap_ref;
    short open_read, ext_read, open_ref, ext_ref;
    int strip_rows;           // > 0: the read was swept in row strips of this many padded rows (strip_kernels.hip.h),
    long long strip_words;    //      each with its own region of the pointer stream, this many dwords apart
    int wide_score;           // 1: int32 cells -- the end value is EndCell.score (low half) and EndCell.pad (high half)
    int *min_start;           // not null: receives the smallest readStart of the launch (atomicMin; the caller presets R + F) --
                              // the host then copies only the columns from there on out of every row (Engine::align_host)
};

typedef unsigned __attribute__((aligned(1))) u32_any_align;   // global dword access at any byte address

// The walk of ONE pair (one lane): from the end cell back along the stored pointers, writing the two right-justified
// gapped rows and the four coordinates.  A chain of dependent loads, so what it costs is memory transactions:
// pointer words are fetched 16 bytes (4 rows x 8 columns) at a time, read / ref bases 4 at a time, and the two
// output rows are written as dwords.  `ptr_pair` is the first word of the pair's lane group in the pointer stream
// (block 0), `ptr_words` the words from there to the end of the group's last block; `half` selects pair A / B of
// the group.  Pointers may be global or LDS (generic): traceback_kernel walks the HBM scratch, the fused small-batch
// kernel (align_fill_tag_kernel<..., FUSED>) the copy it keeps in LDS.
template <bool BYTE_ROWS = false>      // BYTE_ROWS: everything the walk touches sits in LDS (fused kernel) -- explicit LDS reads,
                                       // one byte store per step (no dword at an odd LDS address)
__device__ __forceinline__ int trace_walk(const TraceArgs &a, const unsigned *ptr_pair, long long ptr_words, int half,
                                          const EndCell e, const uint8_t *read, const uint8_t *ref,
                                          uint8_t *row_read, uint8_t *row_ref, short *out) {
    const int R = a.R, F = a.F, AL = R + F, K = a.K;
    const int wpb = (a.affine && a.tagged != 2) ? 2 * K : K;      // words per lane and block of steps
    const int half_shift = half * 16;

    int i = e.read_pos, j = e.ref_pos;
    int h = a.wide_score ? (int)((unsigned)(unsigned short)e.score | ((unsigned)(unsigned short)e.pad << 16)) : (int)e.score;
    int k = AL - 2;
    long long cached_at = -1;              // first word index held in c0..c3 (multiple of 4)
    unsigned c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    int rd_at = -1, rf_at = -1;            // index / 4 of the cached read / ref dwords
    unsigned rd_w = 0, rf_w = 0;
    unsigned out_r = 0, out_f = 0;         // up to 4 pending output bytes per row, newest in the low byte
    int pending = 0;
    int state = 0;                         // affine only: 0 at H, 1 inside F (gap in the ref), 2 inside E

    auto base_at = [](const uint8_t *seq, int len, int pos, int &at, unsigned &w) -> unsigned {
        if ((pos >> 2) != at) {
            at = pos >> 2;
            const int b = at * 4;
            if (BYTE_ROWS) {            // LDS copy of the sequence, dword aligned and padded to a dword: a plain ds_read
                w = *(lds_cu32 *)(lds_offset(seq) + b);
            } else if (b + 4 <= len) {
                w = *reinterpret_cast<const u32_any_align *>(seq + b);
            } else {
                w = 0;
                for (int x = 0; b + x < len; ++x) w |= (unsigned)seq[b + x] << (8 * x);
            }
        }
        return (w >> (8 * (pos & 3))) & 0xFFu;
    };
    // 2-bit code of cell (i, j) from word `wi` of this pair-of-pairs, through a 4-word cache
    auto code_at = [&](long long wi, int t) -> int {
        const long long wb = wi & ~3ll;
        if (wb != cached_at) {
            cached_at = wb;
            if (BYTE_ROWS) {            // the stream sits in LDS, with other LDS of the block behind it: always the 16-byte read
                const u32x4 v = *(lds_cu32x4 *)(lds_offset(ptr_pair) + 4 * (unsigned)wb);
                c0 = v.x; c1 = v.y; c2 = v.z; c3 = v.w;
            } else if (wb + 4 <= ptr_words) {
                const uint4 v = *reinterpret_cast<const uint4 *>(ptr_pair + wb);
                c0 = v.x; c1 = v.y; c2 = v.z; c3 = v.w;
            } else {
                c0 = ptr_pair[wb];
                c1 = wb + 1 < ptr_words ? ptr_pair[wb + 1] : 0u;
                c2 = wb + 2 < ptr_words ? ptr_pair[wb + 2] : 0u;
                c3 = 0u;
            }
        }
        const int sel = (int)(wi & 3);
        const unsigned word = sel == 0 ? c0 : (sel == 1 ? c1 : (sel == 2 ? c2 : c3));
        if (a.tagged == 2) return (int)((word >> (half_shift + 4 * (3 - (t & 3)))) & 15u);
        return (int)((word >> (half_shift + 2 * (7 - (t & 7)))) & 3u);
    };

    while (k >= 0) {
        if (state == 0) {
            if (a.sse_policy) {
                if (i < 0 || (a.alg == kAlgSW && j < 0)) break;      // row 0 (and SW column 0): START
            } else if (a.alg == kAlgSW) {
                if (h <= 0 || i < 0 || j < 0) break;    // cell == 0: START
            } else {
                if (i < 0) break;                       // row 0: START
            }
        }
        int move;                                       // 0 DIAG, 1 UP (emit read, '-'), 2 LEFT
        if (j < 0) {
            move = 1;                                   // column 0 of the NW variant: UP all the way
            state = 0;
        } else {
            int p = i + a.pad_rows;
            long long region = 0;
            if (a.strip_rows > 0) {                     // row strips: each has its own pointer region
                const int strip = p / a.strip_rows;
                p -= strip * a.strip_rows;
                region = strip * a.strip_words;
            }
            const int l = p / K, q = p - l * K;
            const int t = j + l;
            const long long wi = region + ((long long)(a.tagged == 2 ? (t >> 2) : (t >> 3)) * kWave + l) * wpb + q;
            if (a.tagged == 2) {
                const int f4 = code_at(wi, t);          // [3:2] source of H (2 DIAG, 1 F, 0 E), [1] E opened, [0] F opened
                if (state == 0) {
                    const int src = f4 >> 2;
                    move = src == 2 ? 0 : (src == 1 ? 1 : 2);
                    if (move != 0) {
                        state = move;                   // nothing is emitted on entering a gap state
                        continue;
                    }
                } else {
                    move = state;
                    if (state == 1) {
                        if (f4 & 1) { h -= a.open_ref; state = 0; } else h -= a.ext_ref;
                    } else {
                        if (f4 & 2) { h -= a.open_read; state = 0; } else h -= a.ext_read;
                    }
                }
            } else if (a.sse_policy) {
                const int st = code_at(wi, t);          // 0 START, 1 UP, 2 LEFT, 3 DIAG
                if (st == 0) break;
                move = st == 3 ? 0 : st;
            } else if (!a.affine) {
                move = code_at(wi, t);
                if (a.tagged) move = 2 - move;
            } else if (state == 0) {
                move = code_at(wi, t);                  // 0 DIAG, 1 enter F, 2 enter E
                if (move != 0) {
                    state = move;                       // nothing is emitted on entering a gap state
                    continue;
                }
            } else {
                const int bits = code_at(wi + K, t);    // bit0: F extended, bit1: E extended
                move = state;
                if (state == 1) {
                    if (bits & 1) h -= a.ext_ref; else { h -= a.open_ref; state = 0; }
                } else {
                    if (bits & 2) h -= a.ext_read; else { h -= a.open_read; state = 0; }
                }
            }
        }
        unsigned br, bf;
        if (move == 0) {
            br = base_at(read, R, i, rd_at, rd_w);
            bf = base_at(ref, F, j, rf_at, rf_w);
            const int ca = base_class(br), cb = base_class(bf);
            if (ca >= 1 && ca <= 4 && cb >= 1 && cb <= 4) h -= (ca == cb ? a.match : a.mismatch);
            --i;
            --j;
        } else if (move == 1) {
            br = base_at(read, R, i, rd_at, rd_w);
            bf = '-';
            if (!a.affine) h -= a.gap_ref;
            --i;
        } else {
            br = '-';
            bf = base_at(ref, F, j, rf_at, rf_w);
            if (!a.affine) h -= a.gap_read;
            --j;
        }
        if constexpr (BYTE_ROWS) {
            row_read[k] = (uint8_t)br;
            row_ref[k] = (uint8_t)bf;
        } else {
            out_r = (out_r << 8) | br;
            out_f = (out_f << 8) | bf;
            if (++pending == 4) {                       // bytes k .. k+3 of both rows, lowest address = newest
                *reinterpret_cast<u32_any_align *>(row_read + k) = out_r;
                *reinterpret_cast<u32_any_align *>(row_ref + k) = out_f;
                pending = 0;
            }
        }
        --k;
    }
    for (int x = 0; x < pending; ++x) {                 // k + 1 is the newest byte written
        row_read[k + 1 + x] = (uint8_t)(out_r >> (8 * x));
        row_ref[k + 1 + x] = (uint8_t)(out_f >> (8 * x));
    }

    out[0] = (short)(k + 1);
    out[1] = (short)(AL - 1);
    out[2] = (short)(k + 1);
    out[3] = (short)(AL - 1);
    return k + 1;                                       // readStart = refStart: where both rows' strings begin
}

#ifdef VALIGN_TU_ALIGN      // not a template: defined once, in engine_align.hip
// One lane per pair over the HBM pointer scratch.
__global__ void __launch_bounds__(256)
traceback_kernel(const TraceArgs a) {
    const long long pair = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int R = a.R, F = a.F, AL = R + F, K = a.K, G = a.G;
    int start = AL;
    if (pair < a.n) {
        const int wpb = (a.affine && a.tagged != 2) ? 2 * K : K;
        // pointer stream: [wave][block][lane of the wave][wpb]; this pair's group starts at lane (pair-of-pairs % groups) * G
        const int ppw = 2 * (kWave / G);
        const unsigned *ptr_pair = a.ptr + ((pair / ppw) * a.blocks8 * kWave + ((pair % ppw) >> 1) * G) * (long long)wpb;
        long long ptr_words = ((long long)(a.blocks8 - 1) * kWave + G) * wpb;   // words from there to the end of the group's last block
        if (a.strip_rows > 0) ptr_words += (long long)((R + a.pad_rows) / a.strip_rows - 1) * a.strip_words;   // ... of the last strip
        uint8_t *row_read = a.rows + pair * 2 * AL;
        start = trace_walk(a, ptr_pair, ptr_words, (int)(pair & 1), a.ends[pair], a.reads + pair * R, a.refs + pair * F, row_read,
                           row_read + AL, a.idx + pair * 4);
    }
    if (a.min_start) {                                  // (uniform) the launch's smallest start: one atomic per wave
#pragma unroll
        for (int d = kWave / 2; d >= 1; d >>= 1) {
            const int other = __shfl_xor(start, d, kWave);
            start = other < start ? other : start;
        }
        if ((threadIdx.x & (kWave - 1)) == 0 && start < AL) atomicMin(a.min_start, start);
    }
}

// Result rows are right-justified strings behind zeros (650-byte rows whose strings are ~160 bytes for Smith-Waterman
// alignments of 150 bp reads).  Before a chunk's rows cross PCIe, the columns from the chunk's smallest readStart on (rounded
// down to 64) are packed into dense rows of AL - col bytes -- one linear copy then carries a third of the bytes (a pitched
// hipMemcpy2DAsync of the same window took 0.85 s per 131,072 rows on this stack: profiles/r04_d2h_rows.txt).  One block
// per row; *min_start is the word traceback_kernel left.
struct CompactArgs {
    const uint8_t *rows;      // n_rows * AL
    uint8_t *out;             // n_rows * (AL - col)
    const int *min_start;
    long long n_rows;
    int AL;
};

__device__ __forceinline__ int first_copied_column(int min_start, int AL) {
    int c = min_start < 0 ? 0 : (min_start > AL ? AL : min_start);
    return c & ~63;
}

__global__ void __launch_bounds__(256)
compact_rows_kernel(const CompactArgs a) {
    const int col = first_copied_column(*a.min_start, a.AL), w = a.AL - col;
    const uint8_t *src = a.rows + (long long)blockIdx.x * a.AL + col;
    uint8_t *dst = a.out + (long long)blockIdx.x * w;
    for (int k = 4 * threadIdx.x; k + 4 <= w; k += 4 * 256)
        *reinterpret_cast<u32_any_align *>(dst + k) = *reinterpret_cast<const u32_any_align *>(src + k);
    if ((int)threadIdx.x < (w & 3)) dst[(w & ~3) + threadIdx.x] = src[(w & ~3) + threadIdx.x];
}
#endif

// POINTER STREAM LAYOUT: [wave][block of steps][lane of the wave][W words], W = K (2K for the two-stream affine
// kernel).  What a wave stores for one block of steps -- 64 lanes x W dwords -- is ONE contiguous run (2.5 KB at
// K = 10): neighbouring lanes fill each other's 32-byte sectors and DRAM sees whole pages, where the lane-major
// layout of round 1 ([pair-of-pairs][lane][block][W]) scattered a wave's store over 64 places 5 KB apart (its
// HBM write traffic came out 1.2-1.7x the useful bytes even with four blocks parked in registers per burst).
template <int G, int K, int W>
__device__ __forceinline__ unsigned *pointer_stream_lane(unsigned *base, long long pair0, int blocks, int lane) {
    const long long wave = pair0 / Geo<G, K>::kPairs;
    return base + (wave * blocks * kWave + lane) * (long long)W;
}
// block b of the lane's stream (the stride between a lane's blocks is one wave-wide run)
template <int W>
__device__ __forceinline__ unsigned *pointer_stream_block(unsigned *lane_base, long long block) {
    return lane_base + block * (long long)(kWave * W);
}
// W dwords at an 8-byte aligned address (W is even: K is): 8-byte stores
template <int W>
__device__ __forceinline__ void store_block_words(unsigned *dst, const unsigned (&w)[W]) {
    static_assert(W % 2 == 0, "rows per lane are even");
#pragma unroll
    for (int i = 0; i < W / 2; ++i) reinterpret_cast<uint2 *>(dst)[i] = make_uint2(w[2 * i], w[2 * i + 1]);
}

// One block of steps of a lane is complete: store its K words (wave-contiguous layout above).
template <int K>
__device__ __forceinline__ void finish_block(unsigned *lane_base, long long block, const s16x2 (&acc)[K]) {
    unsigned w[K];
#pragma unroll
    for (int q = 0; q < K; ++q) w[q] = as_u32(acc[q]);
    store_block_words<K>(pointer_stream_block<K>(lane_base, block), w);
}

// End cell of each of the two pairs of a lane group, from the per-row first arg-max registers
// (reference rules: DefaultKernel.cpp:252-256 for SW, :307-315 / :381-387 for the NW variant).
template <int G, int K, int ALG, int KEYBITS = 0, int NT = 1, int KEYLOW = 0>     // KEYLOW: bits below the row key (PROFKEY: the tag)
__device__ __forceinline__ void write_end_cells(const FillArgs &args, const WaveTables &w, const s16x2 (&rb)[NT],
                                                const s16x2 (&fc)[NT], const int (&ir)[2], const int (&jr)[2],
                                                int pad_rows, int lane, int grp, int l, int score_shift = 0,
                                                EndCell *wave_ends = nullptr) {     // fused kernel: the wave's own table
    // ---- end cell of each of the two pairs of this group ----
    const int base_lane = lane - l;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const long long pair = w.pair0 + 2 * grp + half;
        EndCell out;
        out.pad = 0;
        if constexpr (ALG == kAlgSW) {
            int bv = 0, bq = 0, bcol = 0;
            if constexpr (KEYBITS > 0) {
                // one (value, row) key per lane: value << KEYBITS | (2^KEYBITS - 1 - row), and its first column
                const int key = half ? rb[0].y : rb[0].x;
                bv = key >> (KEYBITS + KEYLOW);
                bq = ((1 << KEYBITS) - 1) - ((key >> KEYLOW) & ((1 << KEYBITS) - 1));
                bcol = (half ? fc[0].y : fc[0].x) & 0xFFFF;
            } else {
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
            }
            // larger value first, then smaller row: rows are unique per lane so keys are too
            unsigned key = ((unsigned)bv << 16) | (unsigned)(0xFFFF - (l * K + bq));
            unsigned kmax = key;
#pragma unroll
            for (int dd = G / 2; dd >= 1; dd >>= 1) {
                const unsigned other = (unsigned)__shfl_xor((int)kmax, dd, kWave);
                kmax = other > kmax ? other : kmax;
            }
            const int p = 0xFFFF - (int)(kmax & 0xFFFF);
            const int win_lane = p / K;
            const int col_t = __shfl(bcol, base_lane + win_lane, kWave);
            out.score = (short)((kmax >> 16) >> score_shift);     // cells scaled by 4 in the tagged kernel
            out.read_pos = (short)(p - pad_rows);
            out.ref_pos = (short)(col_t - win_lane);
            if (out.score <= 0) {
                out.read_pos = 0;
                out.ref_pos = 0;
            }
        } else {
            const int i_end = ir[half] - 1;               // last valid read position (may be -1)
            int arg_col = 0;
            if (i_end >= 0) {
                const int src_l = (i_end + pad_rows) / K;
                const int mine = ((half ? fc[0].y : fc[0].x) & 0xFFFF) - l;
                arg_col = __shfl(mine, base_lane + src_l, kWave);
            }
            const int last_ref = jr[half] - 1;
            out.score = 0;
            out.read_pos = (short)i_end;
            out.ref_pos = (short)(last_ref < arg_col ? last_ref : arg_col);
        }
        if (l == 0 && pair < args.n) {
            if (wave_ends) wave_ends[2 * grp + half] = out;
            else args.ends[pair] = out;
        }
    }
}

// SYM: gap_read == gap_ref, so one subtract serves both gap directions (as in score_kernel).
template <int G, int K, int ALG, bool SYM>
__global__ void __launch_bounds__(256)
align_fill_kernel(const FillArgs args) {
    using geo = Geo<G, K>;
    const int lane = threadIdx.x & (kWave - 1);
    const int grp = lane / G;
    const int l = lane % G;
    const int R = args.R;
    const int pad_rows = geo::kRows - R;

    WaveTables w;
    if (!wave_setup<G, K, true>(args.reads, args.refs, args.n, R, args.F, args.prof_area, args.refc_stride,
                                args.wave_lds, args.match, args.mismatch, w))
        return;
    // SW: the trailing columns where no reference of the wave has an ACGT base are not swept (they
    // cannot hold the first maximum and no traceback enters them); NW variant: every column.
    const int F = (ALG == kAlgSW) ? w.cols_used : args.F;

    const unsigned lmask = l == 0 ? 0u : 0xFFFFFFFFu;
    const unsigned lane_base = lds_offset(w.prof) + l * geo::kLaneBytes;
    unsigned code_addr = lds_offset(w.refc) + grp * args.refc_stride - 2 * l;

    const s16x2 g_read = pk(ALG == kAlgSW ? (short)-args.gap_read : args.gap_read);
    const s16x2 g_ref = pk(ALG == kAlgSW ? (short)-args.gap_ref : args.gap_ref);
    // Constants of the packed pointer / arg-max arithmetic live in VGPRs the optimiser cannot see
    // through: otherwise min(x, 1), x * 4 and x >> 15 are rewritten into per-half compares and
    // selects (SDWA), which costs twice the instructions of the packed forms.
    s16x2 one = pk(1), four = pk(4), fifteen = pk(15);
    asm volatile("" : "+v"(one), "+v"(four), "+v"(fifteen));

    // NW variant: only ONE row per pair needs its first arg-max -- the row of the last valid read
    // base (DefaultKernel.cpp:307-315, 381-387).  sel[q] marks it per half; the owner lane tracks it.
    int ir[2], jr[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        int p_local = 2 * grp + half;
        p_local = p_local > w.last ? w.last : p_local;
        ir[half] = w.first_bad[2 * p_local];
        jr[half] = w.first_bad[2 * p_local + 1];
    }

    s16x2 Hl[K], code[K], acc[K];
    s16x2 rb[ALG == kAlgSW ? K : 1], fc[ALG == kAlgSW ? K : 1], sel[ALG == kAlgNW ? K : 1];
    short nw_seed[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < K; ++q) {
        const int p = l * K + q;
        short border = 0;
        if (ALG == kAlgNW)                         // column 0 of the NW variant: i * gap_ref, i 1-based
            border = p < pad_rows ? (short)0 : (short)((p - pad_rows + 1) * args.gap_ref);
        Hl[q] = pk(border);
        code[q] = pk(0);
        acc[q] = pk(0);
        if (ALG == kAlgSW) {
            rb[q] = pk(0);
            fc[q] = pk(0);
        } else {
            const bool ta = ir[0] >= 1 && p == ir[0] - 1 + pad_rows;
            const bool tb = ir[1] >= 1 && p == ir[1] - 1 + pad_rows;
            sel[q] = s16x2{(short)(ta ? 1 : 0), (short)(tb ? 1 : 0)};
            if (ta) nw_seed[0] = border;
            if (tb) nw_seed[1] = border;
        }
    }
    if (ALG == kAlgNW) {
        rb[0] = s16x2{nw_seed[0], nw_seed[1]};     // seed of the arg-max: the column-0 value of the row
        fc[0] = pk((short)l);                      // "no cell beat the seed": column index 0
    }
    s16x2 h_last = Hl[K - 1];
    s16x2 up0 = pk(0);
    int j = -l;

    unsigned *ptr_lane = pointer_stream_lane<G, K, K>(args.ptr, w.pair0, args.blocks8, lane);

    auto step = [&](auto masked_tag, int t) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const s16x2 diag0 = up0;
        if (G == 16) {
            up0 = as_pk((unsigned)__builtin_amdgcn_update_dpp(0, (int)as_u32(h_last), 0x111, 0xF, 0xF, true));
        } else {
            up0 = as_pk(group_prev_or_zero<G>(as_u32(h_last), lmask));
        }
        if (!MASKED || (unsigned)j < (unsigned)F) {
            const unsigned ca = *(lds_cu8 *)(code_addr), cb = *(lds_cu8 *)(code_addr + 1);
            s16x2 S[K];
            fetch_profile<G, K>(lane_base + ca * geo::kPairStride, lane_base + cb * geo::kPairStride, S);
            const s16x2 tt = pk((short)t);
            s16x2 d[K], m[K];
#pragma unroll
            for (int q = 0; q < K; ++q) {
                d[q] = (q == 0 ? diag0 : Hl[q - 1]) + S[q];
                if (!SYM) {
                    const s16x2 e = (ALG == kAlgSW) ? pk_sub_floor0(Hl[q], g_read) : Hl[q] + g_read;
                    m[q] = pk_max(d[q], e);
                }
            }
            s16x2 h = up0;
            s16x2 hs = pk(0);
            // The chain down the column (max, sub, max per row) is strictly dependent; the pointer and
            // arg-max arithmetic of the PREVIOUS row is written between its links so that every
            // dependent pair of packed instructions has independent work in between.
            s16x2 nu_prev = pk(0), h_prev = pk(0);
            auto finish_row = [&](int q, s16x2 hq, s16x2 nuq) __attribute__((always_inline)) {
                // back pointer: 0 if h == diag + S, else 1 if it came from above, else 2
                const s16x2 nd = pk_min_u(hq - d[q], one);
                code[q] = (s16x2)((u16x2)nd << (u16x2)nuq);       // nd * (1 + nu): 0 DIAG, 1 UP, 2 LEFT
                if (ALG == kAlgSW) {
                    // per-row first arg-max (strictly greater wins, so the first column is kept);
                    // SW cells are >= 0, so rb - h cannot wrap
                    const s16x2 changed = (rb[q] - hq) >> fifteen;   // 0xFFFF where h beats the row best
                    fc[q] = as_pk((as_u32(changed) & as_u32(tt)) | (~as_u32(changed) & as_u32(fc[q])));
                    rb[q] = pk_max(rb[q], hq);
                } else {
                    unsigned v = as_u32(pk_mad_u(hq, sel[q], hs));      // sel is 1 for the one tracked row: picks its cell
                    asm volatile("" : "+v"(v));                        // (pinned, see align_fill_tag_kernel)
                    hs = as_pk(v);
                }
            };
#pragma unroll
            for (int q = 0; q < K; ++q) {
                s16x2 nu;
                if (SYM) {
                    // h = max(d, max(left, up) - g); it came from UP iff up >= left (priority UP > LEFT)
                    const s16x2 x = pk_max(Hl[q], h);
                    nu = pk_min_u(x - h, one);
                    const s16x2 y = (ALG == kAlgSW) ? pk_sub_floor0(x, g_ref) : x + g_ref;
                    if (q > 0) finish_row(q - 1, h_prev, nu_prev);
                    h = pk_max(d[q], y);
                } else {
                    const s16x2 ug = (ALG == kAlgSW) ? pk_sub_floor0(h, g_ref) : h + g_ref;
                    if (q > 0) finish_row(q - 1, h_prev, nu_prev);
                    h = pk_max(m[q], ug);
                    nu = pk_min_u(h - ug, one);
                }
                Hl[q] = h;
                h_prev = h;
                nu_prev = nu;
            }
            finish_row(K - 1, h_prev, nu_prev);
            if (ALG == kAlgNW) {
                const s16x2 nb = pk_max(rb[0], hs);
                const s16x2 changed = (rb[0] - nb) >> fifteen;
                fc[0] = as_pk((as_u32(changed) & as_u32(tt)) | (~as_u32(changed) & as_u32(fc[0])));
                rb[0] = nb;
            }
            h_last = h;
        }
        // pointer accumulators shift every step in every lane, so that bit positions depend on
        // t only; columns outside [0, F) leave don't-care bits that are never read back
#pragma unroll
        for (int q = 0; q < K; ++q) acc[q] = pk_mad_u(acc[q], four, code[q]);
        if ((t & 7) == 7) {
            unsigned w8[K];
#pragma unroll
            for (int q = 0; q < K; ++q) w8[q] = as_u32(acc[q]);
            store_block_words<K>(pointer_stream_block<K>(ptr_lane, t >> 3), w8);
        }
        ++j;
        code_addr += 2;
    };

    const int steps = (ALG == kAlgSW) ? ((F + G - 1 + 7) / 8) * 8 : args.blocks8 * 8;   // whole 8-step blocks
    const int fill_end = G - 1 < steps ? G - 1 : steps;
    const int steady_end = F > fill_end ? F : fill_end;
    int t = 0;
    for (; t < fill_end; ++t) step(std::true_type{}, t);
    for (; t + 1 < steady_end; t += 2) {       // two steps per trip (loop-carried registers swap roles)
        step(std::false_type{}, t);
        step(std::false_type{}, t + 1);
    }
    for (; t < steady_end; ++t) step(std::false_type{}, t);
    for (; t < steps; ++t) step(std::true_type{}, t);

    write_end_cells<G, K, ALG>(args, w, rb, fc, ir, jr, pad_rows, lane, grp, l);
}

// Linear gaps, Default tie-breaks, with the back pointer carried INSIDE the cell value: every value
// is kept as 4 * H + tag, tag 2 for the diagonal candidate (folded into the query profile: 4 * S + 2),
// 1 for the candidate from above (folded into the gap constant), 0 for the one from the left.  One
// packed maximum then resolves value AND origin with the reference's priority DIAG > UP > LEFT on
// ties; `tag = h & 3` is the pointer, `h - tag` the clean cell for the next column.  Two 32-bit ANDs
// replace the five packed instructions that derive the pointer from equality tests in
// align_fill_kernel: 9 instead of 12 per register.  Needs 4x headroom in int16 and, for SW,
// gap_ref < 0 (the engine falls back to align_fill_kernel otherwise).
// LANEKEY (SW): the reference's end cell is the row-major first cell holding the maximum.  Instead of a
// first-arg-max per row (4 packed instructions per register and step) each lane keeps ONE key
// `value << b | (2^b - 1 - row)` -- larger value first, then the earlier row -- and the step at which
// it last grew: a multiply-add and a maximum per register plus four instructions per step.  Needs
// value << b to stay in int16 (the engine checks).
// SSE: the tie-breaks of the reference's SSE2/AVX2 kernels instead (SSEKernel.cpp:366-379, 646-659): DIAG only
// between two ACGT bases > LEFT > UP, else START, and no "cell == 0 -> START" rule.  The tags ARE the stored
// states then -- 3 on a valid diagonal (profile 4 * S + 3), 0 on an invalid one (profile 0), 2 on the
// candidate from the left, 1 on the one from above, 0 on the SW floor -- cells are computed in the signed form
// (the floor is an explicit maximum) and N counts as invalid for the NW end cell.
// FUSED (small batches, one wave per block): the pointer stream stays in LDS, and the wave that filled it walks its
// own pairs back right away and hands the finished rows out with coalesced stores -- ONE launch per call instead
// of memset + fill + traceback (+ copy): what a small call costs is operations in the stream, ~25 us each, not
// cells (the reference's benchmark is 100 such calls back to back, src/impl/main.cpp:278-287).  The sequences and
// the result rows of the wave's pairs sit in LDS too, so the walk never waits for HBM (or, on the direct path,
// for PCIe).  LDS per wave: fused_lds<G, K>().
template <int G, int K>
struct FusedLds {
    int ptr, reads, refs, rows, ends, total;       // byte offsets behind the wave's tables
};
template <int G, int K>
__host__ __device__ inline FusedLds<G, K> fused_lds(int wave_lds, int R, int F, int blocks8) {
    using geo = Geo<G, K>;
    auto up16 = [](int v) { return (v + 15) / 16 * 16; };
    FusedLds<G, K> f;
    f.ptr = up16(wave_lds);
    const int ptr_end = f.ptr + up16(blocks8 * kWave * K * 4);
    const int reads_bytes = up16(geo::kPairs * ((R + 3) & ~3));    // each sequence starts at a dword: the walk loads bases four at a time
    const int refs_bytes = up16(geo::kPairs * ((F + 3) & ~3));
    const int rows_bytes = up16(geo::kPairs * 2 * (R + F) + 4);
    // What the walk needs beside the pointer stream -- the sequences and the result rows of the wave's pairs -- is staged once
    // the fill is over and the end cells are written: the query profile and the reference codes (the wave's tables, the
    // first `wave_lds` bytes) are dead by then, so the walk's buffers lie over them where they fit.  At 150 x 500 on 64 x 4
    // that takes the block from 82.6 KB to 78.7 KB: two blocks per CU instead of one, and a call of 1,000 pairs (500 blocks
    // on 256 CUs) is one round instead of two.
    if (reads_bytes + refs_bytes + rows_bytes <= f.ptr) {
        f.reads = 0;
        f.refs = reads_bytes;
        f.rows = reads_bytes + refs_bytes;
        f.ends = ptr_end;
    } else {
        f.reads = ptr_end;
        f.refs = f.reads + reads_bytes;
        f.rows = f.refs + refs_bytes;
        f.ends = f.rows + rows_bytes;
    }
    f.total = f.ends + up16(geo::kPairs * (int)sizeof(EndCell));
    return f;
}

// PROFKEY (Smith-Waterman, default tie-breaks, K <= 16): cells are 64 * H + 4 * key + tag, and the lane key of a cell --
// value first, then 15 - row -- is not computed: the query profile carries 4 * (15 - row) in every score of the row, so the
// diagonal candidate IS the key (a maximum of a Smith-Waterman matrix is always a diagonal arrival), and one packed
// maximum per register tracks it.  One multiply-add per register less than LANEKEY; needs 64x headroom in int16.
template <int G, int K, int ALG, bool LANEKEY, bool SSE, bool FUSED = false, bool PROFKEY = false>
__global__ void __launch_bounds__(256)
align_fill_tag_kernel(const FillArgs args) {
    static_assert(!LANEKEY || ALG == kAlgSW, "the lane key replaces the Smith-Waterman row arg-max");
    static_assert(!FUSED || (!LANEKEY && !SSE), "the fused kernel exists for the default tie-breaks");
    static_assert(!PROFKEY || (LANEKEY && !SSE && !FUSED), "the profile key is a form of the lane key");
    constexpr int kScale = PROFKEY ? 64 : 4;
    constexpr int kKeyBits = K <= 16 ? 4 : 5;
    using geo = Geo<G, K>;
    const int lane = threadIdx.x & (kWave - 1);
    const int grp = lane / G;
    const int l = lane % G;
    const int R = args.R;
    const int pad_rows = geo::kRows - R;

    WaveTables w;
    constexpr int kDiagTag = SSE ? 3 : 2;
    // NW, default tie-breaks: the tilted frame of score_kernel -- cell (p, j) plus -gap_ref * p - gap_read * j (times 4) --
    // in which a gap step costs nothing: LEFT is the previous column's clean register as it stands (tag 0), UP the clean
    // cell of the row above with bit 0 set (one full-rate bit operation instead of a packed add each), the diagonal pays
    // both gap scores through the query profile.  Every candidate of a cell is shifted alike: same pointers.
    constexpr bool TILT = ALG == kAlgNW && !SSE;
    const int tilt_row = TILT ? -4 * args.gap_ref : 0, tilt_col = TILT ? -4 * args.gap_read : 0;
    const int tilt_diag = tilt_row + tilt_col;
    if (!wave_setup<G, K, true>(args.reads, args.refs, args.n, R, args.F, args.prof_area, args.refc_stride,
                                args.wave_lds, (short)(kScale * args.match + kDiagTag + tilt_diag),
                                (short)(kScale * args.mismatch + kDiagTag + tilt_diag), w, SSE,
                                blockIdx.x, (short)((SSE ? 0 : 2) + tilt_diag), kOneSweep, (short)(PROFKEY ? 4 : 0)))
        return;
    const int F = (ALG == kAlgSW) ? w.cols_used : args.F;

    const unsigned lmask = l == 0 ? 0u : 0xFFFFFFFFu;
    const unsigned lane_base = lds_offset(w.prof) + l * geo::kLaneBytes;
    unsigned code_addr = lds_offset(w.refc) + grp * args.refc_stride - 2 * l;

    // SW: magnitudes for the unsigned floor-at-zero subtract (4|g| from the left, 4|g| - 1 from above: the
    // result carries tag 1); NW: signed addends 4g and 4g + 1
    // (SSE: signed addends in both modes, LEFT carries tag 2 and UP tag 1)
    constexpr bool kUnsignedGaps = ALG == kAlgSW && !SSE;
    const s16x2 g_read = pk(kUnsignedGaps ? (short)(-kScale * args.gap_read) : (short)(4 * args.gap_read + (SSE ? 2 : 0)));
    const s16x2 g_ref = pk(kUnsignedGaps ? (short)(-kScale * args.gap_ref - 1) : (short)(4 * args.gap_ref + 1));
    s16x2 four = pk(4), fifteen = pk(15), key_mul = pk((short)(1 << (kKeyBits - 2)));     // cells are 4 * H already
    unsigned tag_mask = 0x00030003u, clean4_mask = 0xFFFCFFFCu, up_mask = 0x00010001u;
    unsigned clean_mask = PROFKEY ? 0xFFC0FFC0u : 0xFFFCFFFCu;          // everything below the value
    asm volatile("" : "+v"(four), "+v"(fifteen), "+v"(tag_mask), "+v"(key_mul), "+v"(clean4_mask), "+v"(up_mask), "+v"(clean_mask));

    int ir[2], jr[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        int p_local = 2 * grp + half;
        p_local = p_local > w.last ? w.last : p_local;
        ir[half] = w.first_bad[2 * p_local];
        jr[half] = w.first_bad[2 * p_local + 1];
    }

    s16x2 Hl[K], tag[K], acc[K];
    constexpr int kTracked = (ALG == kAlgSW && !LANEKEY) ? K : 1;
    s16x2 rb[kTracked], fc[kTracked], sel[ALG == kAlgNW ? K : 1];
    s16x2 row_key[LANEKEY ? K : 1];              // 2^b - 1 - q: the earlier row wins among equal values
    short nw_seed[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < K; ++q) {
        const int p = l * K + q;
        short border = 0;
        if (ALG == kAlgNW)                         // column 0 of the NW variant: i * gap_ref, i 1-based (scaled)
            border = p < pad_rows ? (short)0 : (short)(4 * (p - pad_rows + 1) * args.gap_ref);
        Hl[q] = pk((short)(border + tilt_row * p));
        tag[q] = pk(0);
        acc[q] = pk(0);
        if (ALG == kAlgSW) {
            if (LANEKEY) {
                row_key[q] = pk((short)((1 << kKeyBits) - 1 - q));
                asm volatile("" : "+v"(row_key[q]));
                rb[0] = pk(0);
                fc[0] = pk(0);
            } else {
                rb[q] = pk(0);
                fc[q] = pk(0);
            }
        } else {
            const bool ta = ir[0] >= 1 && p == ir[0] - 1 + pad_rows;
            const bool tb = ir[1] >= 1 && p == ir[1] - 1 + pad_rows;
            sel[q] = s16x2{(short)(ta ? 1 : 0), (short)(tb ? 1 : 0)};
            if (ta) nw_seed[0] = border;
            if (tb) nw_seed[1] = border;
        }
    }
    if (ALG == kAlgNW) {
        rb[0] = s16x2{nw_seed[0], nw_seed[1]};
        fc[0] = pk((short)l);
    }
    s16x2 h_last = Hl[K - 1];
    s16x2 up0 = pk(0);
    int j = -l;
    // tilted frame: the all-zero row above padded row 0 as the group leader sees it (one column on per step), and what
    // the tracked row of each pair adds at this lane's column (taken off before the row arg-max)
    s16x2 top_row = pk(0), top_step = pk(0), sel_tilt = pk(0), tilt_step = pk((short)tilt_col);
    if (TILT) {
        if (l == 0) {
            top_row = pk((short)(-tilt_row + tilt_col));
            top_step = pk((short)tilt_col);
            up0 = pk((short)(-tilt_row));
        }
        sel_tilt = s16x2{(short)(tilt_row * (ir[0] - 1 + pad_rows) + tilt_col * (1 - l)),
                         (short)(tilt_row * (ir[1] - 1 + pad_rows) + tilt_col * (1 - l))};
    }

    // FUSED: the wave's pointer stream lives in LDS behind its tables (same layout, wave 0 of its own little scratch)
    const FusedLds<G, K> fused = fused_lds<G, K>(args.wave_lds, R, args.F, args.blocks8);
    unsigned char *wave_smem = valign_smem + (size_t)(threadIdx.x / kWave) * (FUSED ? fused.total : args.wave_lds);
    unsigned *ptr_lane = FUSED ? reinterpret_cast<unsigned *>(wave_smem + fused.ptr) + lane * K
                               : pointer_stream_lane<G, K, K>(args.ptr, w.pair0, args.blocks8, lane);

    // LDS fetches run one step ahead of the arithmetic, as in score_kernel: raw profile dwords of this
    // step in registers, rows of step t+1 and slab numbers of step t+2 requested now (every lane, every
    // step; the code arrays are padded on both sides)
    unsigned pa[K / 2], pb[K / 2];
    unsigned ca_next, cb_next;
    {
        const unsigned ca = *(lds_cu8 *)(code_addr), cb = *(lds_cu8 *)(code_addr + 1);
        lds_load_lane<K>(lane_base + ca * geo::kPairStride, pa);
        lds_load_lane<K>(lane_base + cb * geo::kPairStride, pb);
        ca_next = *(lds_cu8 *)(code_addr + 2);
        cb_next = *(lds_cu8 *)(code_addr + 3);
    }

    // LAST_ONLY (NW): every pair of the wave tracks its last padded row (full-length reads, the usual case):
    // the arg-max bookkeeping then needs the last register only, not a multiply-add per register
    auto step = [&](auto masked_tag, auto last_only_tag, int t) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        constexpr bool LAST_ONLY = decltype(last_only_tag)::value;
        const s16x2 diag0 = up0;
        if (G == 16) {
            up0 = as_pk((unsigned)__builtin_amdgcn_update_dpp(0, (int)as_u32(h_last), 0x111, 0xF, 0xF, true));
        } else {
            up0 = as_pk(group_prev_or_zero<G>(as_u32(h_last), lmask));
        }
        if (TILT) up0 = as_pk(as_u32(up0) | as_u32(top_row));         // (zero in every lane but the group leader)
        s16x2 S[K];
        merge_profile<K>(pa, pb, S);
        lds_load_lane<K>(lane_base + ca_next * geo::kPairStride, pa);
        lds_load_lane<K>(lane_base + cb_next * geo::kPairStride, pb);
        ca_next = *(lds_cu8 *)(code_addr + 4);
        cb_next = *(lds_cu8 *)(code_addr + 5);
        if (!MASKED || (unsigned)j < (unsigned)F) {
            const s16x2 tt = pk((short)t);
            // pass1(q): diagonal and left candidates of row q and their maximum -- only the previous column
            // is needed, so it is computed one row ahead, between the links of the dependent chain
            s16x2 step_key = pk(0);
            auto pass1 = [&](int q) __attribute__((always_inline)) -> s16x2 {
                const s16x2 d = (q == 0 ? diag0 : Hl[q - 1]) + S[q];                               // tag 2 (PROFKEY: and the row key)
                const s16x2 e = TILT ? Hl[q] : (kUnsignedGaps ? pk_sub_floor0(Hl[q], g_read) : Hl[q] + g_read);      // tag 0 (SSE: 2)
                if (PROFKEY) step_key = pk_max(step_key, d);
                return pk_max(d, e);
            };
            s16x2 h = up0;
            s16x2 f_tilt = as_pk(as_u32(up0) | up_mask);           // TILT: the clean cell above, tagged UP
            s16x2 hs = pk(0);
            s16x2 h_prev = pk(0);
            // arg-max bookkeeping of the previous row sits between the links of the dependent chain
            auto finish_row = [&](int q, s16x2 hq) __attribute__((always_inline)) {
                if (PROFKEY) {
                } else if (ALG == kAlgSW && LANEKEY) {
                    step_key = pk_max(step_key, pk_mad_u(hq, key_mul, row_key[q]));
                } else if (ALG == kAlgSW) {
                    const s16x2 changed = (rb[q] - hq) >> fifteen;   // 0xFFFF where h beats the row best
                    fc[q] = as_pk((as_u32(changed) & as_u32(tt)) | (~as_u32(changed) & as_u32(fc[q])));
                    rb[q] = pk_max(rb[q], hq);
                } else {
                    // sel[q] is 1 in the half whose tracked row this is (one row per pair), else 0: a packed
                    // multiply-add picks that row's cell.  Pinned here: sunk to the end of the step pair (where
                    // the optimiser wants it) every clean cell is rematerialised for it.
                    if (!LAST_ONLY || q == K - 1) {
                        unsigned v = as_u32(pk_mad_u(hq, sel[q], hs));
                        asm volatile("" : "+v"(v));
                        hs = as_pk(v);
                    }
                }
            };
            s16x2 m_cur = pass1(0);
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const s16x2 f = TILT ? f_tilt : (kUnsignedGaps ? pk_sub_floor0(h, g_ref) : h + g_ref);                 // tag 1
                s16x2 m_next = pk(0);
                if (q + 1 < K) m_next = pass1(q + 1);          // before Hl[q] is overwritten
                s16x2 ht = pk_max(m_cur, f);
                if (SSE && ALG == kAlgSW) ht = pk_max(ht, pk(0));              // the floor is START (tag 0)
                if (q > 0) finish_row(q - 1, h_prev);
                tag[q] = as_pk(as_u32(ht) & tag_mask);
                if (TILT) f_tilt = as_pk(__builtin_amdgcn_bitop3_b32(as_u32(ht), clean4_mask, up_mask, 0xEA));   // (ht & ~3) | 1
                h = as_pk(as_u32(ht) & clean_mask);
                Hl[q] = h;
                h_prev = h;
                m_cur = m_next;
                // pin the order: left alone, the scheduler sinks the bookkeeping to the end of the step pair
                // and rematerialises every clean cell for it
                __builtin_amdgcn_sched_barrier(0);
            }
            finish_row(K - 1, h_prev);
            if (ALG == kAlgSW && LANEKEY) {
                const s16x2 changed = (rb[0] - step_key) >> fifteen;      // keys are >= 0: no wrap
                fc[0] = as_pk((as_u32(changed) & as_u32(tt)) | (~as_u32(changed) & as_u32(fc[0])));
                rb[0] = pk_max(rb[0], step_key);
            }
            if (ALG == kAlgNW) {
                const s16x2 nb = pk_max(rb[0], TILT ? hs - sel_tilt : hs);    // the tracked row's cell, out of the frame
                const s16x2 changed = (rb[0] - nb) >> fifteen;
                fc[0] = as_pk((as_u32(changed) & as_u32(tt)) | (~as_u32(changed) & as_u32(fc[0])));
                rb[0] = nb;
            }
            h_last = h;
        }
#pragma unroll
        for (int q = 0; q < K; ++q) acc[q] = pk_mad_u(acc[q], four, tag[q]);
        if ((t & 7) == 7) finish_block<K>(ptr_lane, t >> 3, acc);
        if (TILT) {
            top_row = top_row + top_step;
            sel_tilt = sel_tilt + tilt_step;
        }
        ++j;
        code_addr += 2;
    };

    const int steps = (ALG == kAlgSW) ? ((F + G - 1 + 7) / 8) * 8 : args.blocks8 * 8;      // whole blocks
    const int fill_end = G - 1 < steps ? G - 1 : steps;
    const int steady_end = F > fill_end ? F : fill_end;
    auto sweep = [&](auto last_only_tag) __attribute__((always_inline)) {
        int t = 0;
        for (; t < fill_end; ++t) step(std::true_type{}, last_only_tag, t);
        for (; t + 1 < steady_end; t += 2) {
            step(std::false_type{}, last_only_tag, t);
            step(std::false_type{}, last_only_tag, t + 1);
        }
        for (; t < steady_end; ++t) step(std::false_type{}, last_only_tag, t);
        for (; t < steps; ++t) step(std::true_type{}, last_only_tag, t);
    };
    bool last_only = false;
    if (ALG == kAlgNW) {
        bool mine = true;
#pragma unroll
        for (int q = 0; q + 1 < K; ++q) mine = mine && as_u32(sel[q]) == 0u;
        last_only = __all(mine);
    }
    if (last_only) sweep(std::true_type{});
    else sweep(std::false_type{});

    if constexpr (PROFKEY) {
        write_end_cells<G, K, ALG, 4, 1, 2>(args, w, rb, fc, ir, jr, pad_rows, lane, grp, l, 0);
    } else if constexpr (LANEKEY) {
        write_end_cells<G, K, ALG, kKeyBits>(args, w, rb, fc, ir, jr, pad_rows, lane, grp, l, 0);
    } else if constexpr (!FUSED) {
        write_end_cells<G, K, ALG>(args, w, rb, fc, ir, jr, pad_rows, lane, grp, l, 2);
    } else {
        // ---- fused: walk the wave's own pairs back through the LDS pointer stream, then hand the rows out ----
        const int AL = R + args.F;
        EndCell *wave_ends = reinterpret_cast<EndCell *>(wave_smem + fused.ends);
        uint8_t *lds_reads = wave_smem + fused.reads, *lds_refs = wave_smem + fused.refs, *lds_rows = wave_smem + fused.rows;
        write_end_cells<G, K, ALG>(args, w, rb, fc, ir, jr, pad_rows, lane, grp, l, 2, wave_ends);
        // the sequences of the wave's pairs and zeroed result rows (the profile build consumed the staged copy)
        const int pairs_here = w.last + 1;
        const int Rs = (R + 3) & ~3, Fs = (args.F + 3) & ~3;
        for (int x = lane; x < pairs_here * R; x += kWave) lds_reads[(x / R) * Rs + x % R] = args.reads[w.pair0 * R + x];
        for (int x = lane; x < pairs_here * args.F; x += kWave) lds_refs[(x / args.F) * Fs + x % args.F] = args.refs[w.pair0 * args.F + x];
        for (int x = lane; x < (geo::kPairs * 2 * AL + 3) / 4; x += kWave) reinterpret_cast<unsigned *>(lds_rows)[x] = 0u;
        __syncthreads();                       // one wave per block: the LDS writes above are visible to every lane
        short my_idx[4] = {0, 0, 0, 0};
        if (lane < pairs_here) {
            TraceArgs t{};
            t.R = R;
            t.F = args.F;
            t.G = G;
            t.K = K;
            t.pad_rows = pad_rows;
            t.blocks8 = args.blocks8;
            t.alg = ALG;
            t.tagged = 1;
            t.match = args.match;
            t.mismatch = args.mismatch;
            t.gap_read = args.gap_read;
            t.gap_ref = args.gap_ref;
            const unsigned *ptr_pair = reinterpret_cast<const unsigned *>(wave_smem + fused.ptr) + (lane >> 1) * G * K;
            const long long ptr_words = ((long long)(args.blocks8 - 1) * kWave + G) * K;
            uint8_t *row_read = lds_rows + lane * 2 * AL;
            trace_walk<true>(t, ptr_pair, ptr_words, lane & 1, wave_ends[lane], lds_reads + lane * Rs, lds_refs + lane * Fs,
                             row_read, row_read + AL, my_idx);
            short *out = args.out_idx + (w.pair0 + lane) * 4;
            out[0] = my_idx[0];
            out[1] = my_idx[1];
            out[2] = my_idx[2];
            out[3] = my_idx[3];
        }
        __syncthreads();
        // rows of the wave's pairs are one contiguous run of the output: dwords where the address allows, else bytes
        uint8_t *dst = args.out_rows + w.pair0 * 2 * AL;
        const int bytes = pairs_here * 2 * AL;
        if ((reinterpret_cast<unsigned long long>(dst) & 3ull) == 0) {
            for (int x = lane; x < bytes / 4; x += kWave) reinterpret_cast<unsigned *>(dst)[x] = reinterpret_cast<const unsigned *>(lds_rows)[x];
            for (int x = (bytes & ~3) + lane; x < bytes; x += kWave) dst[x] = lds_rows[x];
        } else {
            for (int x = lane; x < bytes; x += kWave) dst[x] = lds_rows[x];
        }
    }
}

// Affine-gap (Gotoh) fill -- an extension, the reference has no affine model.  Per cell two
// 2-bit codes are streamed out: where H came from (0 DIAG, 1 F = gap in the ref, 2 E = gap in the
// read; priority DIAG > F > E so that open == extend walks the linear path), and whether F / E were
// extended (bit set) or opened from H (preferred on ties).  Pointer scratch per lane and 8-step
// block: K dwords of H codes followed by K dwords of gap codes.
// SYM: open_read == open_ref and ext_read == ext_ref -- `H + open` is computed once per cell and serves E of
// the next column and F of the next row (as in score_kernel's kGapAffineSym).
template <int G, int K, int ALG, bool SYM>
__global__ void __launch_bounds__(256)
align_fill_affine_kernel(const FillArgs args) {
    using geo = Geo<G, K>;
    const int lane = threadIdx.x & (kWave - 1);
    const int grp = lane / G;
    const int l = lane % G;
    const int R = args.R, F = args.F;
    const int pad_rows = geo::kRows - R;

    WaveTables w;
    if (!wave_setup<G, K, true>(args.reads, args.refs, args.n, R, F, args.prof_area, args.refc_stride,
                                args.wave_lds, args.match, args.mismatch, w))
        return;

    const unsigned lmask = l == 0 ? 0u : 0xFFFFFFFFu;
    const unsigned lane_base = lds_offset(w.prof) + l * geo::kLaneBytes;
    unsigned code_addr = lds_offset(w.refc) + grp * args.refc_stride - 2 * l;

    s16x2 o_read, e_read, o_ref, e_ref;
    if (ALG == kAlgSW) {
        o_read = pk((short)-args.open_read);  e_read = pk((short)-args.ext_read);
        o_ref = pk((short)-args.open_ref);    e_ref = pk((short)-args.ext_ref);
    } else {
        o_read = pk(args.open_read);  e_read = pk(args.ext_read);
        o_ref = pk(args.open_ref);    e_ref = pk(args.ext_ref);
    }
    const s16x2 border_f = pk(ALG == kAlgNW ? kNegInf : (short)0);
    s16x2 one = pk(1), two = pk(2), four = pk(4), fifteen = pk(15);
    asm volatile("" : "+v"(one), "+v"(two), "+v"(four), "+v"(fifteen));

    int ir[2], jr[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        int p_local = 2 * grp + half;
        p_local = p_local > w.last ? w.last : p_local;
        ir[half] = w.first_bad[2 * p_local];
        jr[half] = w.first_bad[2 * p_local + 1];
    }

    s16x2 Hl[K], El[K], code_h[K], code_g[K], acc_h[K], acc_g[K];
    s16x2 HOl[SYM ? K : 1];                    // H + open of the previous column
    s16x2 rb[ALG == kAlgSW ? K : 1], fc[ALG == kAlgSW ? K : 1], sel[ALG == kAlgNW ? K : 1];
    short nw_seed[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < K; ++q) {
        const int p = l * K + q;
        short border = 0;
        if (ALG == kAlgNW)                         // column 0: a gap of i bases in the ref direction
            border = p < pad_rows ? (short)0 : (short)(args.open_ref + (p - pad_rows) * args.ext_ref);
        Hl[q] = pk(border);
        if (SYM) HOl[q] = (ALG == kAlgSW) ? pk_sub_floor0(Hl[q], o_read) : pk_add_sat(Hl[q], o_read);
        El[q] = border_f;
        code_h[q] = code_g[q] = acc_h[q] = acc_g[q] = pk(0);
        if (ALG == kAlgSW) {
            rb[q] = pk(0);
            fc[q] = pk(0);
        } else {
            const bool ta = ir[0] >= 1 && p == ir[0] - 1 + pad_rows;
            const bool tb = ir[1] >= 1 && p == ir[1] - 1 + pad_rows;
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
    s16x2 fup_keep = border_f;       // NW variant, 16-lane groups: F of the row above the lane's rows (see the step)
    s16x2 up0 = pk(0);
    int j = -l;

    unsigned *ptr_lane = pointer_stream_lane<G, K, 2 * K>(args.ptr, w.pair0, args.blocks8, lane);

    // LDS fetches run one step ahead of the arithmetic, as in score_kernel: raw profile dwords of this
    // step in registers, rows of step t+1 and slab numbers of step t+2 requested now (every lane, every
    // step; the code arrays are padded on both sides)
    unsigned pa[K / 2], pb[K / 2];
    unsigned ca_next, cb_next;
    {
        const unsigned ca = *(lds_cu8 *)(code_addr), cb = *(lds_cu8 *)(code_addr + 1);
        lds_load_lane<K>(lane_base + ca * geo::kPairStride, pa);
        lds_load_lane<K>(lane_base + cb * geo::kPairStride, pb);
        ca_next = *(lds_cu8 *)(code_addr + 2);
        cb_next = *(lds_cu8 *)(code_addr + 3);
    }

    auto step = [&](auto masked_tag, int t) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const s16x2 diag0 = up0;
        up0 = as_pk(group_prev_or_zero<G>(as_u32(h_last), lmask));
        s16x2 fup0;
        if (G == 16 && ALG == kAlgSW) {          // (row_shr:1: the group's first lane reads 0, the Smith-Waterman border)
            fup0 = as_pk((unsigned)__builtin_amdgcn_update_dpp(0, (int)as_u32(f_last), 0x111, 0xF, 0xF, true));
        } else if (G == 16) {                    // NW: the group's first lane keeps what the register held -- the border, for good
            fup_keep = as_pk((unsigned)__builtin_amdgcn_update_dpp((int)as_u32(fup_keep), (int)as_u32(f_last), 0x111, 0xF, 0xF, false));
            fup0 = fup_keep;
        } else {
            const unsigned fv = from_prev_lane(as_u32(f_last));
            fup0 = (ALG == kAlgNW) ? as_pk(l == 0 ? as_u32(border_f) : fv) : as_pk(fv & lmask);
        }
        s16x2 S[K];
        merge_profile<K>(pa, pb, S);
        lds_load_lane<K>(lane_base + ca_next * geo::kPairStride, pa);
        lds_load_lane<K>(lane_base + cb_next * geo::kPairStride, pb);
        ca_next = *(lds_cu8 *)(code_addr + 4);
        cb_next = *(lds_cu8 *)(code_addr + 5);
        if (!MASKED || (unsigned)j < (unsigned)F) {
            const s16x2 tt = pk((short)t);
            s16x2 d[K], m[K];
#pragma unroll
            for (int q = 0; q < K; ++q) {
                d[q] = (q == 0 ? diag0 : Hl[q - 1]) + S[q];
                const s16x2 e_open = SYM ? HOl[q] : ((ALG == kAlgSW) ? pk_sub_floor0(Hl[q], o_read) : pk_add_sat(Hl[q], o_read));
                const s16x2 e_extd = (ALG == kAlgSW) ? pk_sub_floor0(El[q], e_read) : pk_add_sat(El[q], e_read);
                const s16x2 e = pk_max(e_extd, e_open);
                El[q] = e;
                code_g[q] = pk_min_u(e - e_open, one);            // 1: E extended, 0: opened from H
                m[q] = pk_max(d[q], e);
            }
            s16x2 h = up0, f = fup0;
            s16x2 hs = pk(0);
            s16x2 ho = (ALG == kAlgSW) ? pk_sub_floor0(up0, o_ref) : pk_add_sat(up0, o_ref);     // SYM: H + open of the row above
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const s16x2 f_open = SYM ? ho : ((ALG == kAlgSW) ? pk_sub_floor0(h, o_ref) : pk_add_sat(h, o_ref));
                const s16x2 f_extd = (ALG == kAlgSW) ? pk_sub_floor0(f, e_ref) : pk_add_sat(f, e_ref);
                f = pk_max(f_extd, f_open);
                h = pk_max(m[q], f);
                Hl[q] = h;
                if (SYM) {
                    ho = (ALG == kAlgSW) ? pk_sub_floor0(h, o_ref) : pk_add_sat(h, o_ref);
                    HOl[q] = ho;
                }
                const s16x2 nd = pk_min_u(h - d[q], one);
                const s16x2 nf = pk_min_u(h - f, one);
                code_h[q] = (s16x2)((u16x2)nd << (u16x2)nf);      // nd * (1 + nf): 0 DIAG, 1 from F, 2 from E
                code_g[q] = pk_mad_u(code_g[q], two, pk_min_u(f - f_open, one));   // bit1 E extended, bit0 F extended
                if (ALG == kAlgSW) {
                    const s16x2 changed = (rb[q] - h) >> fifteen;
                    fc[q] = as_pk((as_u32(changed) & as_u32(tt)) | (~as_u32(changed) & as_u32(fc[q])));
                    rb[q] = pk_max(rb[q], h);
                } else {
                    unsigned v = as_u32(pk_mad_u(h, sel[q], hs));       // sel is 1 for the one tracked row: picks its cell
                    asm volatile("" : "+v"(v));                        // (pinned, see align_fill_tag_kernel)
                    hs = as_pk(v);
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
#pragma unroll
        for (int q = 0; q < K; ++q) {
            acc_h[q] = pk_mad_u(acc_h[q], four, code_h[q]);
            acc_g[q] = pk_mad_u(acc_g[q], four, code_g[q]);
        }
        if ((t & 7) == 7) {
            unsigned w8[2 * K];
#pragma unroll
            for (int q = 0; q < K; ++q) w8[q] = as_u32(acc_h[q]);
#pragma unroll
            for (int q = 0; q < K; ++q) w8[K + q] = as_u32(acc_g[q]);
            store_block_words<2 * K>(pointer_stream_block<2 * K>(ptr_lane, t >> 3), w8);
        }
        ++j;
        code_addr += 2;
    };

    const int steps = args.blocks8 * 8;
    const int fill_end = G - 1 < steps ? G - 1 : steps;
    const int steady_end = F > fill_end ? F : fill_end;
    int t = 0;
    for (; t < fill_end; ++t) step(std::true_type{}, t);
    for (; t + 1 < steady_end; t += 2) {       // two steps per trip (loop-carried registers swap roles)
        step(std::false_type{}, t);
        step(std::false_type{}, t + 1);
    }
    for (; t < steady_end; ++t) step(std::false_type{}, t);
    for (; t < steps; ++t) step(std::true_type{}, t);

    write_end_cells<G, K, ALG>(args, w, rb, fc, ir, jr, pad_rows, lane, grp, l);
}


// Affine gaps with everything the traceback needs carried in the low bits of the cell values (the
// affine counterpart of align_fill_tag_kernel).  All values are 8 * x + tag:
//   E / F candidates: bit 0 = 1 on "opened from H" (folded into the open constant), 0 on "extended" --
//     one maximum picks the value and, on ties, the open, as the equality-test kernel does;
//   H candidates: bits 2..1 = 2 on the diagonal term (folded into the query profile: 8 * S + 4), 1 on
//     F (one add), 0 on E -- one maximum chain resolves DIAG > F > E on ties whatever bit 0 holds.
// `x & ~7` cleans a value for reuse, and one 4-bit code per cell -- source of H, E opened, F opened --
// is three bit operations and a multiply-add away.  18 packed instructions per register instead of
// 22; the pointer stream is K dwords per lane and 4-step block (same volume as before).
// SYM: open_read == open_ref and ext_read == ext_ref, `H + open` shared by E and F.
// NW (no zero floor) runs in a tilted frame: every value of cell (p, j) is kept as
//   V'(p, j) = V(p, j) - ext_ref * p - ext_read * j        (p padded row, j matrix column; times 8)
// so that extending a gap costs nothing -- E' = max(E'(p, j-1), H'(p, j-1) + open_read - ext_read),
// F' = max(F'(p-1, j), H'(p-1, j) + open_ref - ext_ref) -- and the diagonal term pays both extensions,
// folded into the query profile (S - ext_ref - ext_read).  Two packed adds per register less; the pointers are
// those of the plain frame (every candidate of a cell is shifted by the same amount), the row arg-max of
// the end-cell rule is taken on un-tilted values (two instructions per step).
template <int G, int K, int ALG, bool SYM>
__global__ void __launch_bounds__(256)
align_fill_affine_tag_kernel(const FillArgs args) {
    using geo = Geo<G, K>;
    constexpr bool LANEKEY = ALG == kAlgSW;
    constexpr bool TILT = ALG == kAlgNW;
    constexpr int kKeyBits = K <= 16 ? 4 : 5;
    const int lane = threadIdx.x & (kWave - 1);
    const int grp = lane / G;
    const int l = lane % G;
    const int R = args.R;
    const int pad_rows = geo::kRows - R;

    const int tilt_row = TILT ? -8 * args.ext_ref : 0, tilt_col = TILT ? -8 * args.ext_read : 0;
    const int tilt_diag = tilt_row + tilt_col;
    WaveTables w;
    if (!wave_setup<G, K, true>(args.reads, args.refs, args.n, R, args.F, args.prof_area, args.refc_stride,
                                args.wave_lds, (short)(8 * args.match + 4 + tilt_diag),
                                (short)(8 * args.mismatch + 4 + tilt_diag), w, false, blockIdx.x, (short)(4 + tilt_diag)))
        return;
    const int F = (ALG == kAlgSW) ? w.cols_used : args.F;

    const unsigned lmask = l == 0 ? 0u : 0xFFFFFFFFu;
    const unsigned lane_base = lds_offset(w.prof) + l * geo::kLaneBytes;
    unsigned code_addr = lds_offset(w.refc) + grp * args.refc_stride - 2 * l;

    // SW: magnitudes for the unsigned floor-at-zero subtract (the open constants one short: the result carries
    // tag 1); NW: signed saturating addends
    s16x2 x_read, x_ref, o_read, o_ref;
    if (ALG == kAlgSW) {
        x_read = pk((short)(-8 * args.ext_read));      x_ref = pk((short)(-8 * args.ext_ref));
        o_read = pk((short)(-8 * args.open_read - 1)); o_ref = pk((short)(-8 * args.open_ref - 1));
    } else {                                           // tilted frame: extensions are free, opens cost open - extend
        x_read = x_ref = pk(0);
        o_read = pk((short)(8 * (args.open_read - args.ext_read) + 1));
        o_ref = pk((short)(8 * (args.open_ref - args.ext_ref) + 1));
    }
    constexpr short kMinusInf = -30000;                // multiple of 8, far below any real cell, room to saturate
    const s16x2 border_f = pk(ALG == kAlgNW ? kMinusInf : (short)0);
    s16x2 two = pk(2), sixteen = pk(16), fifteen = pk(15), key_mul = pk((short)(1 << (kKeyBits - 3)));
    unsigned clean_mask = 0xFFF8FFF8u, src_mask = 0x00060006u, one_mask = 0x00010001u, two_mask = 0x00020002u;
    asm volatile("" : "+v"(two), "+v"(sixteen), "+v"(fifteen), "+v"(key_mul), "+v"(clean_mask), "+v"(src_mask), "+v"(one_mask), "+v"(two_mask));

    auto gap_add = [](s16x2 v, s16x2 c) __attribute__((always_inline)) {
        return (ALG == kAlgSW) ? pk_sub_floor0(v, c) : pk_add_sat(v, c);
    };

    int ir[2], jr[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        int p_local = 2 * grp + half;
        p_local = p_local > w.last ? w.last : p_local;
        ir[half] = w.first_bad[2 * p_local];
        jr[half] = w.first_bad[2 * p_local + 1];
    }

    s16x2 Hl[K], El[K], HOl[SYM ? K : 1], code[K], acc[K];
    s16x2 rb[1], fc[1], sel[ALG == kAlgNW ? K : 1], row_key[LANEKEY ? K : 1];
    short nw_seed[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < K; ++q) {
        const int p = l * K + q;
        short border = 0;
        if (ALG == kAlgNW)                         // column 0: a gap of i bases in the ref direction (times 8)
            border = p < pad_rows ? (short)0 : (short)(8 * (args.open_ref + (p - pad_rows) * args.ext_ref));
        Hl[q] = pk((short)(border + tilt_row * p));
        if (SYM) HOl[q] = gap_add(Hl[q], o_read);
        El[q] = border_f;
        code[q] = acc[q] = pk(0);
        if (ALG == kAlgSW) {
            row_key[q] = pk((short)((1 << kKeyBits) - 1 - q));
            asm volatile("" : "+v"(row_key[q]));
        } else {
            const bool ta = ir[0] >= 1 && p == ir[0] - 1 + pad_rows;
            const bool tb = ir[1] >= 1 && p == ir[1] - 1 + pad_rows;
            sel[q] = s16x2{(short)(ta ? 1 : 0), (short)(tb ? 1 : 0)};
            if (ta) nw_seed[0] = border;
            if (tb) nw_seed[1] = border;
        }
    }
    rb[0] = pk(0);
    fc[0] = pk(0);
    if (ALG == kAlgNW) {
        rb[0] = s16x2{nw_seed[0], nw_seed[1]};
        fc[0] = pk((short)l);
    }
    s16x2 h_last = Hl[K - 1], f_last = border_f;
    s16x2 fup_keep = border_f;       // NW variant, 16-lane groups: F of the row above the lane's rows (see the step)
    s16x2 up0 = pk(0);
    int j = -l;
    // tilted frame: the all-zero row above padded row 0 reads -ext_ref * (-1) - ext_read * j in lane 0 of a group,
    // and the tracked row's cell is un-tilted before the arg-max (per pair: its own row)
    s16x2 top_row = pk(0), top_step = pk(0), sel_tilt = pk(0), tilt_step = pk((short)tilt_col);
    if (TILT) {
        if (l == 0) {
            top_row = pk((short)(-tilt_row + tilt_col));           // column 1 at step 0
            top_step = pk((short)tilt_col);
            up0 = pk((short)(-tilt_row));                           // column 0: the diagonal of the first cell
        }
        sel_tilt = s16x2{(short)(tilt_row * (ir[0] - 1 + pad_rows) + tilt_col * (1 - l)),
                         (short)(tilt_row * (ir[1] - 1 + pad_rows) + tilt_col * (1 - l))};
    }

    unsigned *ptr_lane = pointer_stream_lane<G, K, K>(args.ptr, w.pair0, args.blocks8, lane);     // blocks8: 4-step blocks here

    unsigned pa[K / 2], pb[K / 2];
    unsigned ca_next, cb_next;
    {
        const unsigned ca = *(lds_cu8 *)(code_addr), cb = *(lds_cu8 *)(code_addr + 1);
        lds_load_lane<K>(lane_base + ca * geo::kPairStride, pa);
        lds_load_lane<K>(lane_base + cb * geo::kPairStride, pb);
        ca_next = *(lds_cu8 *)(code_addr + 2);
        cb_next = *(lds_cu8 *)(code_addr + 3);
    }

    // LAST_ONLY (NW): every pair of the wave tracks its last padded row (full-length reads, the usual case):
    // the arg-max bookkeeping then needs the last register only, not a multiply-add per register
    auto step = [&](auto masked_tag, auto last_only_tag, int t) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        constexpr bool LAST_ONLY = decltype(last_only_tag)::value;
        const s16x2 diag0 = up0;
        up0 = TILT ? as_pk(group_prev_or_zero<G>(as_u32(h_last), lmask) | as_u32(top_row))
                   : as_pk(group_prev_or_zero<G>(as_u32(h_last), lmask));
        s16x2 fup0;
        if (G == 16 && ALG == kAlgSW) {          // (row_shr:1: the group's first lane reads 0, the Smith-Waterman border)
            fup0 = as_pk((unsigned)__builtin_amdgcn_update_dpp(0, (int)as_u32(f_last), 0x111, 0xF, 0xF, true));
        } else if (G == 16) {                    // NW: the group's first lane keeps what the register held -- the border, for good
            fup_keep = as_pk((unsigned)__builtin_amdgcn_update_dpp((int)as_u32(fup_keep), (int)as_u32(f_last), 0x111, 0xF, 0xF, false));
            fup0 = fup_keep;
        } else {
            const unsigned fv = from_prev_lane(as_u32(f_last));
            fup0 = (ALG == kAlgNW) ? as_pk(l == 0 ? as_u32(border_f) : fv) : as_pk(fv & lmask);
        }
        s16x2 S[K];
        merge_profile<K>(pa, pb, S);
        lds_load_lane<K>(lane_base + ca_next * geo::kPairStride, pa);
        lds_load_lane<K>(lane_base + cb_next * geo::kPairStride, pb);
        ca_next = *(lds_cu8 *)(code_addr + 4);
        cb_next = *(lds_cu8 *)(code_addr + 5);
        if (!MASKED || (unsigned)j < (unsigned)F) {
            const s16x2 tt = pk((short)t);
            // pass1(q): everything of row q that only needs the previous column, one row ahead of the chain
            s16x2 d_cur, e_cur;
            auto pass1 = [&](int q, s16x2 &d_t, s16x2 &e_t) __attribute__((always_inline)) {
                d_t = (q == 0 ? diag0 : Hl[q - 1]) + S[q];                                     // tag 4: DIAG
                const s16x2 e_ext = TILT ? El[q] : gap_add(El[q], x_read);                     // bit 0 = 0: extended
                const s16x2 e_opn = SYM ? HOl[q] : gap_add(Hl[q], o_read);                     // bit 0 = 1: opened
                e_t = pk_max(e_ext, e_opn);
                El[q] = as_pk(as_u32(e_t) & clean_mask);
            };
            s16x2 hc = up0, fcl = fup0;                      // clean H / F of the row above
            s16x2 ho = gap_add(up0, o_ref);                  // its H + open (bit 0 = 1)
            s16x2 hs = pk(0), step_key = pk(0);
            pass1(0, d_cur, e_cur);
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const s16x2 f_ext = TILT ? fcl : gap_add(fcl, x_ref);
                const s16x2 f_t = pk_max(f_ext, ho);
                s16x2 d_next = pk(0), e_next = pk(0);
                if (q + 1 < K) pass1(q + 1, d_next, e_next);                 // before Hl[q] is overwritten
                fcl = as_pk(as_u32(f_t) & clean_mask);
                // The bit operations below are single 32-bit instructions over both halves on purpose: plain AND / OR /
                // ADD and v_bitop3_b32 issue at twice the rate of the packed-16 and the other three-operand integer
                // forms (tools/microbench/valu_rate2.hip); none of them carries across the halves.
                const s16x2 f_h = as_pk(as_u32(f_t) | two_mask);              // bits 2..1 = 1: from F (bit 1 is clear in F)
                const s16x2 h_t = pk_max(pk_max(d_cur, f_h), e_cur);
                hc = as_pk(as_u32(h_t) & clean_mask);
                Hl[q] = hc;
                ho = gap_add(hc, o_ref);
                if (SYM) HOl[q] = ho;
                // 4-bit code: [3:2] source of H (2 DIAG, 1 F, 0 E), [1] E opened, [0] F opened
                const unsigned src_e = __builtin_amdgcn_bitop3_b32(as_u32(h_t), src_mask, as_u32(e_cur) & one_mask, 0xEA);    // (h & 6) | e0
                unsigned src_e2;
                asm("v_add_u32 %0, %1, %1" : "=v"(src_e2) : "v"(src_e));      // (the compiler would pick the half-rate shift)
                code[q] = as_pk(__builtin_amdgcn_bitop3_b32(as_u32(f_t), one_mask, src_e2, 0xEA));                           // 2 * that | f0
                if (ALG == kAlgSW) {
                    step_key = pk_max(step_key, pk_mad_u(hc, key_mul, row_key[q]));
                } else {
                    if (!LAST_ONLY || q == K - 1) {
                        unsigned v = as_u32(pk_mad_u(hc, sel[q], hs));
                        asm volatile("" : "+v"(v));
                        hs = as_pk(v);
                    }
                }
                d_cur = d_next;
                e_cur = e_next;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ALG == kAlgSW) {
                const s16x2 changed = (rb[0] - step_key) >> fifteen;
                fc[0] = as_pk((as_u32(changed) & as_u32(tt)) | (~as_u32(changed) & as_u32(fc[0])));
                rb[0] = pk_max(rb[0], step_key);
            } else {
                const s16x2 nb = pk_max(rb[0], hs - sel_tilt);                // the tracked row's cell, un-tilted
                const s16x2 changed = (rb[0] - nb) >> fifteen;
                fc[0] = as_pk((as_u32(changed) & as_u32(tt)) | (~as_u32(changed) & as_u32(fc[0])));
                rb[0] = nb;
            }
            h_last = hc;
            f_last = fcl;
        }
#pragma unroll
        for (int q = 0; q < K; ++q) acc[q] = pk_mad_u(acc[q], sixteen, code[q]);
        if ((t & 3) == 3) finish_block<K>(ptr_lane, t >> 2, acc);
        if (TILT) {
            top_row = top_row + top_step;
            sel_tilt = sel_tilt + tilt_step;
        }
        ++j;
        code_addr += 2;
    };

    const int steps = (ALG == kAlgSW) ? ((F + G - 1 + 3) / 4) * 4 : args.blocks8 * 4;      // whole blocks
    const int fill_end = G - 1 < steps ? G - 1 : steps;
    const int steady_end = F > fill_end ? F : fill_end;
    auto sweep = [&](auto last_only_tag) __attribute__((always_inline)) {
        int t = 0;
        for (; t < fill_end; ++t) step(std::true_type{}, last_only_tag, t);
        for (; t + 1 < steady_end; t += 2) {
            step(std::false_type{}, last_only_tag, t);
            step(std::false_type{}, last_only_tag, t + 1);
        }
        for (; t < steady_end; ++t) step(std::false_type{}, last_only_tag, t);
        for (; t < steps; ++t) step(std::true_type{}, last_only_tag, t);
    };
    bool last_only = false;
    if (ALG == kAlgNW) {
        bool mine = true;
#pragma unroll
        for (int q = 0; q + 1 < K; ++q) mine = mine && as_u32(sel[q]) == 0u;
        last_only = __all(mine);
    }
    if (last_only) sweep(std::true_type{});
    else sweep(std::false_type{});

    if constexpr (LANEKEY) write_end_cells<G, K, ALG, kKeyBits>(args, w, rb, fc, ir, jr, pad_rows, lane, grp, l, 0);
    else write_end_cells<G, K, ALG>(args, w, rb, fc, ir, jr, pad_rows, lane, grp, l, 0);
}

// Second tie-break policy: the reference's SSE2/AVX2 kernels (SSEKernel.cpp:366-379, 646-659).
// Stored states: 0 START, 1 UP, 2 LEFT, 3 DIAG; DIAG only between two ACGT bases, then LEFT, then UP,
// else START -- no "cell == 0 -> START" rule.  Cells are computed in the signed form (the equality
// tests are against the un-floored up+gap / left+gap), so the zero floor is an explicit max.
template <int G, int K, int ALG>
__global__ void __launch_bounds__(256)
align_fill_sse_kernel(const FillArgs args) {
    using geo = Geo<G, K>;
    const int lane = threadIdx.x & (kWave - 1);
    const int grp = lane / G;
    const int l = lane % G;
    const int R = args.R, F = args.F;
    const int pad_rows = geo::kRows - R;

    WaveTables w;
    if (!wave_setup<G, K, true>(args.reads, args.refs, args.n, R, F, args.prof_area, args.refc_stride,
                                args.wave_lds, args.match, args.mismatch, w, true))
        return;

    const unsigned lmask = l == 0 ? 0u : 0xFFFFFFFFu;
    const unsigned lane_base = lds_offset(w.prof) + l * geo::kLaneBytes;
    unsigned code_addr = lds_offset(w.refc) + grp * args.refc_stride - 2 * l;

    const s16x2 g_read = pk(args.gap_read), g_ref = pk(args.gap_ref);
    s16x2 one = pk(1), three = pk(3), four = pk(4), fifteen = pk(15);
    asm volatile("" : "+v"(one), "+v"(three), "+v"(four), "+v"(fifteen));

    int ir[2], jr[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        int p_local = 2 * grp + half;
        p_local = p_local > w.last ? w.last : p_local;
        ir[half] = w.first_bad[2 * p_local];
        jr[half] = w.first_bad[2 * p_local + 1];
    }

    s16x2 Hl[K], code[K], acc[K], rinv[K];
    s16x2 rb[ALG == kAlgSW ? K : 1], fc[ALG == kAlgSW ? K : 1], sel[ALG == kAlgNW ? K : 1];
    short nw_seed[2] = {0, 0};
    {
        int pa = 2 * grp, pb = 2 * grp + 1;
        pa = pa > w.last ? w.last : pa;
        pb = pb > w.last ? w.last : pb;
        const uint8_t *ra = args.reads + (w.pair0 + pa) * R, *rbp = args.reads + (w.pair0 + pb) * R;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const int p = l * K + q;
            const int grow = p - pad_rows;
            const int ca = grow >= 0 ? base_class(ra[grow]) : 0, cb = grow >= 0 ? base_class(rbp[grow]) : 0;
            rinv[q] = s16x2{(short)((ca >= 1 && ca <= 4) ? 0 : 1), (short)((cb >= 1 && cb <= 4) ? 0 : 1)};
            short border = 0;
            if (ALG == kAlgNW) border = p < pad_rows ? (short)0 : (short)((p - pad_rows + 1) * args.gap_ref);
            Hl[q] = pk(border);
            code[q] = pk(0);
            acc[q] = pk(0);
            if (ALG == kAlgSW) {
                rb[q] = pk(0);
                fc[q] = pk(0);
            } else {
                const bool ta = ir[0] >= 1 && p == ir[0] - 1 + pad_rows;
                const bool tb = ir[1] >= 1 && p == ir[1] - 1 + pad_rows;
                sel[q] = s16x2{(short)(ta ? -1 : 0), (short)(tb ? -1 : 0)};
                if (ta) nw_seed[0] = border;
                if (tb) nw_seed[1] = border;
            }
        }
    }
    if (ALG == kAlgNW) {
        rb[0] = s16x2{nw_seed[0], nw_seed[1]};
        fc[0] = pk((short)l);
    }
    s16x2 h_last = Hl[K - 1];
    s16x2 up0 = pk(0);
    int j = -l;

    unsigned *ptr_lane = pointer_stream_lane<G, K, K>(args.ptr, w.pair0, args.blocks8, lane);

    auto step = [&](auto masked_tag, int t) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const s16x2 diag0 = up0;
        up0 = as_pk(group_prev_or_zero<G>(as_u32(h_last), lmask));
        if (!MASKED || (unsigned)j < (unsigned)F) {
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
                lg[q] = Hl[q] + g_read;
            }
            s16x2 h = up0;
            s16x2 hs = pk(0);
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const s16x2 ug = h + g_ref;
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
                    hs = as_pk((as_u32(sel[q]) & as_u32(h)) | (~as_u32(sel[q]) & as_u32(hs)));
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
        if ((t & 7) == 7) {
            unsigned w8[K];
#pragma unroll
            for (int q = 0; q < K; ++q) w8[q] = as_u32(acc[q]);
            store_block_words<K>(pointer_stream_block<K>(ptr_lane, t >> 3), w8);
        }
        ++j;
        code_addr += 2;
    };

    const int steps = args.blocks8 * 8;
    const int fill_end = G - 1 < steps ? G - 1 : steps;
    const int steady_end = F > fill_end ? F : fill_end;
    int t = 0;
    for (; t < fill_end; ++t) step(std::true_type{}, t);
    for (; t + 1 < steady_end; t += 2) {       // two steps per trip (loop-carried registers swap roles)
        step(std::false_type{}, t);
        step(std::false_type{}, t + 1);
    }
    for (; t < steady_end; ++t) step(std::false_type{}, t);
    for (; t < steps; ++t) step(std::true_type{}, t);

    write_end_cells<G, K, ALG>(args, w, rb, fc, ir, jr, pad_rows, lane, grp, l);
}


}  // namespace valign
