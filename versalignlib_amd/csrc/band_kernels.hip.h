// band_kernels.hip.h -- banded Smith-Waterman scores of long reads (BASELINE config 5: 10 kbp x 10 kbp, 512
// diagonals, int32 cells) as a CYCLIC systolic chain.  Same recurrence and results as score_long_kernel
// (reference semantics: src/Kernels/default/DefaultKernel.cpp:83-138 on the cells of the band), different schedule:
//
//   * A lane group is G = 32 lanes; lane l owns ROW BLOCK b = m * G + l (K rows) of "strip" m and sweeps ONLY that
//     block's own band window [lo_b, hi_b] -- the band of include/valign_hip.h on blocks of K rows -- one column per
//     step.  score_long_kernel sweeps, per 160-row strip, the rectangle that contains the band of all its rows:
//     at 512 diagonals a quarter of its lane-steps lie outside the band and every strip pays a pipeline fill.
//   * Block b starts d steps after block b - 1 (d: the window width divided by G, rounded up, so that a lane is done
//     with block b when block b + G is due).  The bottom row of block b - 1 reaches block b through an LDS DELAY
//     RING: every lane writes its last row's cell each step, its successor reads the cell written D_b = d -
//     (start_b - start_{b-1}) steps earlier -- the same column.  Lane 0 of strip m + 1 follows lane G - 1 of strip m
//     through the very same ring: the chain is a cycle, NO strip boundary row ever goes to HBM (score_long_kernel:
//     15 GB per 32,768 pairs for 0.66 GB of sequences) and no lane waits for a strip to drain.
//   * A lane is "inactive" outside its window (EXEC-masked; it writes 0 into its ring, which is what the band gives
//     the cells out there) -- blocks of top padding, the warm-up column that fetches the diagonal neighbour of a
//     block's first cell, and the d * G - width idle steps of a period.
//   * Every d steps is an EVENT (wave-uniform): block b = t / d starts on lane b % G of both groups.  The event
//     resets that lane's cells, loads its window, rewrites ITS rows of the query profile in place (all 64 lanes
//     cooperate; the read bases were requested one event earlier), and tops up the group's reference ring
//     (requested one event earlier as well).  No barrier anywhere: one wave per block.
//   * int32 cells (one pair per register: the group's two pairs take turns, as in score_long_kernel).  Round 3 read an
//     int32 query profile (no sign extension per cell); round 4 packs it as int16, two rows per dword, laid out
//     [8 rows][slab][lane] so that a lane's K scores are K/8 ds_read_b128 at 16-byte lane stride (conflict-free within a
//     slab) -- the add that forms the diagonal candidate takes its score straight out of the register half
//     (v_add_u32_sdwa ... sext WORD_0 / WORD_1: same instruction count).  What it buys is LDS: 18.7 -> 9.5 KB per wave,
//     sixteen one-wave blocks per CU instead of eight -- the kernel is bound by how often a WAVE may issue, not by
//     dependency latency (profiles/r04_band_two_chains.txt), so only more waves fill the SIMD's issue slots.
//
// tools/band_schedule_model.py states this schedule in plain Python and checks it against the oracle's block band;
// the constants of the band definition are reported by valign_hip_describe ("band_block_rows", "band_col_align").
#pragma once

#include "dp_kernels.hip.h"

namespace valign {

constexpr int kBandG = 32;                 // lanes per group: two groups (= two pairs at a time) per wave
constexpr int kBandGroups = kWave / kBandG;
constexpr int kBandSlabs = 4 * kBandGroups + 1;                // (class, group) + one all-zero slab (both groups)

struct BandBlock {          // per row block, computed once on the host (Engine::band_plan)
    int start;              // column of the block's first step (its window's first column - 1: the warm-up column)
    int lo;                 // first column of the window, clipped to the matrix (0x3FFFFFFF: nothing to compute)
    int span;               // last column - first column
    int delay;              // steps between the predecessor's write and this block's read of the same column
};

struct BandArgs {
    const uint8_t *reads;
    const uint8_t *refs;
    int16_t *scores;
    const BandBlock *blocks;    // nb + kBandG + 2 entries (the tail: empty blocks)
    const int *fill_to;         // per event: the reference ring must hold every column below this one
    long long n;
    int R, F;
    int nb;                     // blocks = strips * kBandG
    int first_block;            // the first block that holds a row of the read (the ones before it are top padding: skipped)
    int pad_rows;               // padding rows above the read (class "none")
    int d;                      // steps between block starts
    int ring_depth;             // delay ring slots per lane (power of two > every delay)
    int code_cols;              // reference ring columns per group (power of two)
    short match, mismatch;
    short gap_read, gap_ref;
    short open_read, ext_read, open_ref, ext_ref;     // AFFINE instantiations (all <= 0)
};

template <int K>
struct BandLds {
    static constexpr int kRowChunks = K / 8;
    static constexpr int kChunkBytes = kBandSlabs * kBandG * 16;          // one 8-row chunk (int16 scores) of every slab
    static constexpr int kProf = 0;
    static constexpr int kProfBytes = kRowChunks * kChunkBytes;
    // (the reference rings are addressed as base | column: aligned to their own size, code_cols a power of two)
    __host__ __device__ static int codes(int code_cols) { return (kProfBytes + code_cols - 1) / code_cols * code_cols; }
    // (the delay rings are addressed as base | offset: aligned to one lane's ring)
    __host__ __device__ static int ring(int code_cols, int ring_depth) {
        const int at = codes(code_cols) + kBandGroups * code_cols, a = ring_depth * 4;
        return (at + a - 1) / a * a;
    }
    // (ring_depth 0: the unit-delay kernel, no ring; affine: a second ring, for F, behind the first)
    __host__ __device__ static int total(int code_cols, int ring_depth, bool affine = false) {
        return ring(code_cols, ring_depth ? ring_depth : 1) + kWave * ring_depth * 4 * (affine ? 2 : 1);
    }
};

// SYM: gap_read == gap_ref (one saturating subtract per cell serves both neighbours).
// UNIT: every block reads its predecessor's cell of the step before (delay 1: the window starts advance by d - 1 columns
// per block, BASELINE config 5's 10 kbp x 10 kbp at 512 diagonals is such a case) -- then the cell travels by DPP
// (lane 0 <- lane 31, lane 32 <- lane 63 through two scalar registers) and no LDS round trip sits between two steps of the
// chain.  Otherwise the delay ring, whose reads run one step ahead (the host plans every delay >= 2 for that).
// AFFINE (round 4): Gotoh's recurrence on the same chain -- E lives in registers like H (one value per row, carried along
// the row), F runs down the column: through the lane's rows in registers and from lane to lane beside H (a second DPP
// move / a second delay ring).  All values floored at 0 by the saturating subtracts, which for Smith-Waterman is the
// recurrence itself (a gap score below zero never beats the floor).  SYM then means open_read == open_ref and ext_read ==
// ext_ref: H - open is computed once per cell and serves E of the next column and F of the next row.
// 3-bit class codes for the letters 'A' + t (t = 0..19): A 0, T 1, C 2, G 3 -- the class order of the profile's slabs -- every
// other letter and entry 20 (what all other bytes clamp to): 4
constexpr unsigned long long band_class_table() {
    unsigned long long table = 0;
    for (int t = 0; t <= 20; ++t) {
        const unsigned long long c = t == 0 ? 0 : (t == 19 ? 1 : (t == 2 ? 2 : (t == 6 ? 3 : 4)));
        table |= c << (3 * t);
    }
    return table;
}
static_assert(((band_class_table() >> (3 * ('A' - 'A'))) & 7) == 0 && ((band_class_table() >> (3 * ('T' - 'A'))) & 7) == 1 &&
                  ((band_class_table() >> (3 * ('C' - 'A'))) & 7) == 2 && ((band_class_table() >> (3 * ('G' - 'A'))) & 7) == 3 &&
                  ((band_class_table() >> (3 * ('N' - 'A'))) & 7) == 4 && ((band_class_table() >> (3 * 20)) & 7) == 4,
              "classes of the profile's slabs: A, T, C, G; N and everything else: none");

template <int K, bool SYM, bool UNIT, bool AFFINE = false>
__global__ void __launch_bounds__(64)
score_band_kernel(const BandArgs args) {
    static_assert(K % 8 == 0, "rows per lane come in chunks of eight int16 scores");
    using lay = BandLds<K>;
    const int lane = threadIdx.x;
    const int grp = lane / kBandG;
    const int l = lane % kBandG;
    const int R = args.R, F = args.F, d = args.d;
    constexpr int kPairsPerWave = 2 * kBandGroups;

    unsigned char *codes = valign_smem + lay::codes(args.code_cols);
    const unsigned prof_lds = lds_offset(valign_smem);
    const unsigned ring_lds = lds_offset(valign_smem + lay::ring(args.code_cols, UNIT ? 1 : args.ring_depth));
    // delay ring, slot-major: slot s of lane x at ring + s * 256 + x * 4 (a step's 64 stores hit 64 banks)
    const unsigned ring_mask = ((unsigned)args.ring_depth - 1u) * 256u;          // of 256 * slot
    const unsigned code_mask = (unsigned)args.code_cols - 1u;
    const unsigned my_ring = ring_lds + (unsigned)lane * 4u;
    const unsigned pred_ring = ring_lds + (unsigned)(grp * kBandG + ((l + kBandG - 1) % kBandG)) * 4u;
    const unsigned codes_lds = lds_offset(codes) + (unsigned)grp * (unsigned)args.code_cols;      // aligned to code_cols (<= 2048)
    // a lane's scores of class slab s: K/8 chunks of 16 bytes (8 int16) at prof + chunk * kChunkBytes + (s * G + l) * 16
    const unsigned lane_prof = prof_lds + (unsigned)l * 16u;
    constexpr unsigned kSlabStride = kBandG * 16;
    constexpr unsigned zero_slab = 4 * kBandGroups;                   // (class * kBandGroups + grp for a real class)
    const unsigned gmag_ref = (unsigned)(-(int)args.gap_ref), gmag_read = (unsigned)(-(int)args.gap_read);
    const unsigned omag_read = (unsigned)(-(int)args.open_read), emag_read = (unsigned)(-(int)args.ext_read);
    const unsigned omag_ref = (unsigned)(-(int)args.open_ref), emag_ref = (unsigned)(-(int)args.ext_ref);
    const unsigned ring_f = (unsigned)kWave * (unsigned)args.ring_depth * 4u;       // the F ring sits behind the H ring

    // the all-zero slabs and the never-filled part of the reference ring, once
    for (int i = lane; i < lay::kRowChunks * kBandG * 4; i += kWave) {
        const int chunk = i / (kBandG * 4), rest = i % (kBandG * 4);
        reinterpret_cast<unsigned *>(valign_smem + chunk * lay::kChunkBytes + zero_slab * kSlabStride)[rest] = 0u;
    }
    // (the other slabs: every (lane, row) entry is written by the event that starts the lane's block, before its first use)

  // A launch is as many one-wave blocks as the device runs side by side (the engine asks the occupancy calculator); every
  // wave takes quads of pairs in turn.  With a block per quad the 32 waves a CU gets for 32,768 pairs went through it in
  // 4.7 rounds' time instead of 4 (waves of a round do not retire together and CUs are not handed equal shares).
  for (long long pair0 = (long long)blockIdx.x * kPairsPerWave; pair0 < args.n; pair0 += (long long)gridDim.x * kPairsPerWave) {
  const int last = (int)((args.n - pair0 < kPairsPerWave ? args.n - pair0 : kPairsPerWave) - 1);
  // the group's two pairs take turns (int32 cells: one pair per register)
  for (int half = 0; half < 2; ++half) {
    int p_local = 2 * grp + half;
    p_local = p_local > last ? last : p_local;
    const uint8_t *my_ref = args.refs + (pair0 + p_local) * F;

    for (int i = lane; i < kBandGroups * args.code_cols; i += kWave) codes[i] = (unsigned char)zero_slab;
    if (!UNIT)
        for (int i = lane; i < kWave * args.ring_depth * (AFFINE ? 2 : 1); i += kWave)
            reinterpret_cast<unsigned *>(valign_smem + lay::ring(args.code_cols, args.ring_depth))[i] = 0u;

    // per row: H of the previous column; linear gaps: max(H - g, 0) beside it (shared-gap form); affine: E of the previous
    // column and, with symmetric scores, max(H - open, 0)
    // H of the lane's K rows at its current column and, beside it in a 64-bit register pair, G = H - gap (shared gap score:
    // what both neighbours subtract) -- pairs so that the event's reset of a lane is K 64-bit moves instead of 2 K
    typedef unsigned long long u64;
    u64 HG[K];
    int El[AFFINE ? K : 1];
    auto h_of = [&](int q) __attribute__((always_inline)) -> int { return (int)(unsigned)HG[q]; };
    auto g_of = [&](int q) __attribute__((always_inline)) -> int { return (int)(unsigned)(HG[q] >> 32); };
    auto put_hg = [&](int q, int h, int g) __attribute__((always_inline)) { HG[q] = ((u64)(unsigned)g << 32) | (u64)(unsigned)h; };
    auto put_h = [&](int q, int h) __attribute__((always_inline)) { HG[q] = (HG[q] & 0xFFFFFFFF00000000ull) | (u64)(unsigned)h; };
#pragma unroll
    for (int q = 0; q < K; ++q) HG[q] = 0;
#pragma unroll
    for (int q = 0; q < (AFFINE ? K : 1); ++q) El[q] = 0;
    int up0 = 0, up_in = 0, best = 0;      // up_in: the predecessor's cell for the coming step (read one step ahead)
    int fup_in = 0;                        // affine: the predecessor's F beside it
    // u: the lane's column minus the first column of its window (inside the window while 0 <= u <= span);
    // ca: LDS address of the ring entry two columns ahead of the lane's
    int u = -0x20000000, span = 0;
    unsigned ca = codes_lds;
    unsigned rd4 = 0;                      // 256 * (t + 1 - delay): the predecessor's slot of the column this lane reaches NEXT step
    unsigned t4 = 0;                       // 256 * t

    // ---- what the events prefetch ----
    // Read bases of the NEXT block's rows: this lane rewrites the profile entry (row my_q, class my_c) of group 0 and
    // of group 1 -- 64 lanes x 2 = 16 rows x 4 classes x 2 groups.
    static_assert(K == 16, "the cooperative profile rewrite below maps 64 lanes onto 16 rows x 4 classes");
    const int my_q = (lane >> 2) & (K - 1), my_c = lane & 3;
    const int pg0 = (half > last) ? last : half, pg1 = (2 + half > last) ? last : 2 + half;
    const uint8_t *rows0 = args.reads + (pair0 + pg0) * R - args.pad_rows + my_q;     // + b * K: row my_q of block b
    const uint8_t *rows1 = args.reads + (pair0 + pg1) * R - args.pad_rows + my_q;
    const int row_first = args.pad_rows - my_q;                                 // b * K must lie in [row_first, row_first + R)
    auto row_base = [&](const uint8_t *rows, int b) __attribute__((always_inline)) -> unsigned {
        const int at = b * K;
        return (b < args.nb && (unsigned)(at - row_first) < (unsigned)R) ? (unsigned)rows[at] : 0u;
    };
    // class of a base without a branch: 0..3 for A / T / C / G in either case, 4 for anything else -- 3-bit entries for the
    // letters 'A' + t, t = 0..19, in one 64-bit constant; every other byte clamps to entry 20 (bytes >= 0x80 keep bit 7 under
    // the case fold and bytes below 'A' wrap: both land far beyond 20)
    constexpr unsigned long long kClassTable = band_class_table();
    auto class_of = [](unsigned ch) __attribute__((always_inline)) -> unsigned {
        unsigned t = (ch & 0xDFu) - 'A';
        t = t < 20u ? t : 20u;
        return (unsigned)(kClassTable >> (3u * t)) & 7u;
    };
    // substitution score of a read base against this lane's class: 0 unless both are one of ACGT
    auto entry_score = [&](unsigned ch) __attribute__((always_inline)) -> int {
        const unsigned c = class_of(ch);
        return c == (unsigned)my_c ? (int)args.match : (c < 4u ? (int)args.mismatch : 0);
    };
    // reference bases for the ring: lane -> column first + lane % 32 (+ 32 in the second round) of its group's pair;
    // the ring holds the slab of the lane-independent part of the profile address: class * groups + group, or the zero slab
    const uint8_t *refl = my_ref + l;
    auto ref_base = [&](int first) __attribute__((always_inline)) -> unsigned {
        return first + l < F ? (unsigned)refl[first] : 0u;
    };
    auto slab_of = [&](unsigned ch) __attribute__((always_inline)) -> unsigned {
        const unsigned slab = class_of(ch) * kBandGroups + (unsigned)grp;     // class 4 -> 8 or 9: the zero slab
        return slab < zero_slab ? slab : zero_slab;
    };
    // (a round none of whose lanes has a column to commit is skipped as a whole: the second one at every event that brings at
    // most 32 new columns -- all events of a square matrix)
    auto commit_codes = [&](int first, int limit, unsigned b0, unsigned b1) __attribute__((always_inline)) {
        const int c0 = first + l, c1 = first + kBandG + l;
        if (c0 < limit) codes[grp * args.code_cols + (c0 & code_mask)] = (unsigned char)slab_of(b0);
        if (c1 < limit) codes[grp * args.code_cols + (c1 & code_mask)] = (unsigned char)slab_of(b1);
    };

    // before the first event: the ring up to the first block's fill target (synchronously, once), that block's read bases.
    // The chain starts with the first block that holds a row of the read: the blocks of top padding before it (15 of 640
    // at 10 kbp) would only cost their d steps each.
    const int b0 = args.first_block;
    int filled = 0;
    for (; filled < args.fill_to[b0]; filled += 2 * kBandG)
        commit_codes(filled, args.fill_to[b0], ref_base(filled), ref_base(filled + kBandG));
    filled = args.fill_to[b0];
    unsigned pre_row0 = row_base(rows0, b0), pre_row1 = row_base(rows1, b0);
    unsigned pre_ref0 = 0, pre_ref1 = 0;
    int pre_first = filled, pre_limit = filled;

    // this step's and the next step's scores: K int16, rows 2i / 2i + 1 in the halves of dword i
    u32x4 S0[K / 8], S1[K / 8];
#pragma unroll
    for (int q = 0; q < K / 8; ++q) S0[q] = S1[q] = u32x4{0u, 0u, 0u, 0u};
    // LDS address of the lane's scores for the NEXT step (the slab of its next column; carried as an address: a carried
    // byte makes the compiler re-mask it every step)
    unsigned addr_next = lane_prof + zero_slab * kSlabStride;
    auto load_scores = [&](unsigned addr, u32x4 (&S)[K / 8]) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < K / 8; ++c) S[c] = *(lds_cu32x4 *)(addr + c * lay::kChunkBytes);
    };
    // row q's score out of its register half, sign-extended (folds into the add that uses it)
    auto score_of = [](const u32x4 (&S)[K / 8], int q) __attribute__((always_inline)) -> int {
        const unsigned w = S[q >> 3][(q >> 1) & 3];
        return (q & 1) ? ((int)w >> 16) : (int)(short)(w & 0xFFFFu);
    };

    // one step: every lane moves one column on.  `S` holds this step's scores, the loads of the next step's go to Snext.
    auto step = [&](u32x4 (&S)[K / 8], u32x4 (&Snext)[K / 8]) __attribute__((always_inline)) {
        const int diag0 = up0;
        int fup_cur = 0;
        if (!(UNIT && K == 16)) {
            up0 = up_in;                                                 // the cell above this block's first row
            fup_cur = fup_in;                                            // (affine) ... and its F
            if (!UNIT) {
                up_in = (int)*(lds_cu32 *)(pred_ring + (rd4 & ring_mask));   // ... of the next step (written >= 1 step ago)
                if (AFFINE) fup_in = (int)*(lds_cu32 *)(pred_ring + ring_f + (rd4 & ring_mask));
            }
        }
        // (inline assembly: through the compiler the code byte comes back as an "any-extended" load and is masked again
        // every step; ds_read_u8 zero-extends.  The compiler does not count these loads -- LDS returns in order, its own
        // waits only become more conservative -- so the wait sits with the one use, at the end of the step.)
        unsigned code;
        if constexpr (UNIT && K == 16) {
            // every LDS operation of the step in one place, so that the count behind the hand-over's ds_swizzle (issued at
            // the end of the previous step, see below) is known: two score loads and the code byte later, it has landed
            // (the loads land in the registers the next step reads: no copy of a register in flight)
            if constexpr (AFFINE)
                asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:%7\n\tds_read_u8 %2, %6\n\ts_waitcnt lgkmcnt(3)"
                             : "=&v"(Snext[0]), "=&v"(Snext[1]), "=&v"(code), "+v"(up_in), "+v"(fup_in)
                             : "v"(addr_next), "v"(ca), "n"(lay::kChunkBytes));
            else
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_u8 %2, %5\n\ts_waitcnt lgkmcnt(3)"
                             : "=&v"(Snext[0]), "=&v"(Snext[1]), "=&v"(code), "+v"(up_in)
                             : "v"(addr_next), "v"(ca), "n"(lay::kChunkBytes));
            up0 = up_in;                                                 // (only now: the registers were in flight)
            fup_cur = fup_in;
        } else {
            load_scores(addr_next, Snext);                               // step t + 1
            asm volatile("ds_read_u8 %0, %1" : "=v"(code) : "v"(ca));    // step t + 2
        }
        ca = ((ca + 1u) & code_mask) | codes_lds;
        int h_out = 0, f_out = 0;
        if (AFFINE) {
            if ((unsigned)u <= (unsigned)span) {
                int f = fup_cur;
                int ho = (int)__builtin_elementwise_sub_sat((unsigned)up0, omag_ref);          // H - open of the row above
                int d_cur = diag0 + score_of(S, 0), d_prev = 0, h = 0;
#pragma unroll
                for (int q = 0; q < K; ++q) {
                    int d_next = 0;
                    if (q + 1 < K) d_next = h_of(q) + score_of(S, q + 1);  // before H of the row is overwritten
                    const int ex = (int)__builtin_elementwise_sub_sat((unsigned)El[q], emag_read);
                    const int eo = SYM ? g_of(q) : (int)__builtin_elementwise_sub_sat((unsigned)h_of(q), omag_read);
                    const int e = ex > eo ? ex : eo;
                    const int fx = (int)__builtin_elementwise_sub_sat((unsigned)f, emag_ref);
                    f = fx > ho ? fx : ho;
                    int m = d_cur > e ? d_cur : e;
                    m = m > f ? m : f;
                    h = m;
                    El[q] = e;
                    ho = (int)__builtin_elementwise_sub_sat((unsigned)m, omag_ref);
                    if (SYM) put_hg(q, m, ho); else put_h(q, m);
                    if (q & 1) {
                        int b2 = best > d_prev ? best : d_prev;
                        best = b2 > d_cur ? b2 : d_cur;
                    } else if (q == K - 1) {
                        best = best > d_cur ? best : d_cur;
                    }
                    d_prev = d_cur;
                    d_cur = d_next;
                }
                h_out = h;
                f_out = f;
            }
        } else if ((unsigned)u <= (unsigned)span) {
            int h = up0;
            int up_c = (int)__builtin_elementwise_sub_sat((unsigned)up0, gmag_ref);
            int d_cur = diag0 + score_of(S, 0), d_prev = 0;
#pragma unroll
            for (int q = 0; q < K; ++q) {
                int d_next = 0;
                if (q + 1 < K) d_next = h_of(q) + score_of(S, q + 1);    // before H of the row is overwritten
                int left_c = g_of(q);
                if (!SYM) left_c = (int)__builtin_elementwise_sub_sat((unsigned)h_of(q), gmag_read);
                int m = d_cur > left_c ? d_cur : left_c;
                m = m > up_c ? m : up_c;
                h = m;
                up_c = (int)__builtin_elementwise_sub_sat((unsigned)m, gmag_ref);
                if (SYM) put_hg(q, m, up_c); else put_h(q, m);
                if (q & 1) {
                    int b2 = best > d_prev ? best : d_prev;
                    best = b2 > d_cur ? b2 : d_cur;
                } else if (q == K - 1) {
                    best = best > d_cur ? best : d_cur;
                }
                d_prev = d_cur;
                d_cur = d_next;
            }
            h_out = h;
        }
        if (!UNIT) {
            *(__attribute__((address_space(3))) unsigned *)(my_ring + (t4 & ring_mask)) = (unsigned)h_out;
            if (AFFINE) *(__attribute__((address_space(3))) unsigned *)(my_ring + ring_f + (t4 & ring_mask)) = (unsigned)f_out;
            rd4 += 256;
            t4 += 256;
        }
        static_assert(kSlabStride == 512, "the shift below");
        if constexpr (UNIT && K == 16) {
            // lane l's next cell from above is what lane l - 1 just computed (0 outside its window), lanes 0 and 32 take
            // lanes 31 and 63: the rotation of each half of the wave by one lane is ONE ds_swizzle (rotate mode, no LDS
            // memory, no VALU slot -- the DPP shift with its two readlane / writelane fix-ups was 5 VALU instructions a
            // step), issued BEHIND the step's wait: its latency passes under the next step's loads and the other waves
            if constexpr (AFFINE)
                asm volatile("s_waitcnt lgkmcnt(0)\n\tv_lshl_add_u32 %0, %3, 9, %4\n\t"
                             "ds_swizzle_b32 %1, %5 offset:swizzle(ROTATE,1,1)\n\tds_swizzle_b32 %2, %6 offset:swizzle(ROTATE,1,1)"
                             : "=&v"(addr_next), "=&v"(up_in), "=&v"(fup_in) : "v"(code), "v"(lane_prof), "v"(h_out), "v"(f_out));
            else
                asm volatile("s_waitcnt lgkmcnt(0)\n\tv_lshl_add_u32 %0, %2, 9, %3\n\tds_swizzle_b32 %1, %4 offset:swizzle(ROTATE,1,1)"
                             : "=&v"(addr_next), "=&v"(up_in) : "v"(code), "v"(lane_prof), "v"(h_out));
            ++u;
            return;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\tv_lshl_add_u32 %0, %1, 9, %2" : "=v"(addr_next) : "v"(code), "v"(lane_prof));
        if (UNIT) {
            // lane l's next cell from above is what lane l - 1 just computed (0 outside its window), lanes 0 and 32 take
            // lanes 31 and 63: the rotation of each half of the wave by one lane is ONE ds_swizzle (rotate mode, no LDS
            // memory, no VALU slot -- the DPP shift with its two readlane / writelane fix-ups was 5 VALU instructions a
            // step); issued behind the step's wait, its latency passes under the other waves of the SIMD
            up_in = __builtin_amdgcn_ds_swizzle(h_out, 0xC420 /* swizzle(ROTATE, 1, 1): lane i <- lane i - 1 within 32 */);
            if (AFFINE) fup_in = __builtin_amdgcn_ds_swizzle(f_out, 0xC420);
        }
        ++u;
    };

    // ================= event: block b starts on lane b % G of both groups =================
    // The switching lane's first step is its warm-up column (inactive: the scores in flight for it may be anything);
    // what must be right is the slab of the step after, read here.  Nobody else's pipeline is touched.
    int b = b0;
    auto event = [&]() __attribute__((always_inline)) {
        const int ls = b % kBandG;
        const BandBlock blk = args.blocks[b];                             // (uniform: scalar loads)
        commit_codes(pre_first, pre_limit, pre_ref0, pre_ref1);           // (first: the new block's columns may be among them)
        filled = pre_limit > filled ? pre_limit : filled;
        if (l == ls) {
#pragma unroll
            for (int q = 0; q < K; ++q) HG[q] = 0;
#pragma unroll
            for (int q = 0; q < (AFFINE ? K : 1); ++q) El[q] = 0;
            up0 = 0;
            u = blk.start - blk.lo;             // (an empty block: lo = 0x3FFFFFFF -- never inside)
            span = blk.span;
            addr_next = lane_prof + (unsigned)*(lds_cu8 *)(codes_lds | ((unsigned)(blk.start + 1) & code_mask)) * kSlabStride;
            ca = codes_lds | ((unsigned)(blk.start + 2) & code_mask);
            if (!UNIT) {
                // this step's cell from above now, the read for the next step follows in the step itself
                up_in = (int)*(lds_cu32 *)(pred_ring + ((t4 - 256u * (unsigned)blk.delay) & ring_mask));
                if (AFFINE) fup_in = (int)*(lds_cu32 *)(pred_ring + ring_f + ((t4 - 256u * (unsigned)blk.delay) & ring_mask));
                rd4 = t4 + 256u - 256u * (unsigned)blk.delay;
            }
        }
        // this lane's entries of lane ls's profile rows (bases requested at the previous event)
        {
            const unsigned at = ((unsigned)my_q >> 3) * lay::kChunkBytes + (unsigned)ls * 16u + ((unsigned)my_q & 7u) * 2u +
                                (unsigned)my_c * (kBandGroups * kSlabStride);
            *reinterpret_cast<short *>(valign_smem + at) = (short)entry_score(pre_row0);
            *reinterpret_cast<short *>(valign_smem + at + kSlabStride) = (short)entry_score(pre_row1);
        }
        // requests for the next event (their latency hides behind the d steps in between)
        ++b;
        pre_row0 = row_base(rows0, b);
        pre_row1 = row_base(rows1, b);
        pre_first = filled;
        pre_limit = args.fill_to[b];
        pre_ref0 = ref_base(pre_first);
        pre_ref1 = ref_base(pre_first + kBandG);
    };

    // the last blocks need a full period to finish; two steps per trip (the score registers swap roles), the event
    // due before either of them
    const int total_steps = (args.nb - b0 + kBandG) * d;
    int next_event = 0;
    for (int t = 0; t < total_steps; t += 2) {
        if (t == next_event) {
            event();
            next_event += d;
        }
        step(S0, S1);
        if (t + 1 == next_event) {
            event();
            next_event += d;
        }
        step(S1, S0);
    }

    int res = best;
#pragma unroll
    for (int dd = kBandG / 2; dd >= 1; dd >>= 1) {
        const int other = __shfl_xor(res, dd, kWave);
        res = other > res ? other : res;
    }
    if (l == 0) {
        const long long pa = pair0 + 2 * grp + half;
        if (pa < args.n) args.scores[pa] = (int16_t)(res > 32767 ? 32767 : res);     // the ABI's score is a short
    }
  }
  }
}

}  // namespace valign
