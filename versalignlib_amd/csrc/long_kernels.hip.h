// long_kernels.hip.h -- score kernel for sequences that do not fit one register sweep
// (BASELINE config 5: 10 kbp x 10 kbp).  Same recurrences and results as score_kernel
// (reference semantics: src/Kernels/default/DefaultKernel.cpp:83-202), different staging:
//
//   * ROW STRIPS.  The read is cut into strips of G*K rows (160 for the 16x10 geometry).  A lane
//     group sweeps one strip over the whole reference, then the next; the bottom row of a strip
//     (one packed dword per column and pair-of-pairs) goes to an HBM row buffer and comes back
//     as the "row above" of the next strip.  Padding rows sit above strip 0, as in score_kernel.
//   * COLUMN PHASES.  Nothing of length F lives in LDS: every kPhase (64) steps the wave refills
//     three small LDS rings -- the next 64 reference class codes of its 8 pairs, the next 64
//     boundary values coming in, and it drains the 64 boundary values going out -- all with
//     coalesced 16-byte accesses.  LDS per wave is ~17 KB whatever R and F are.
//   * The query profile is rebuilt per strip (2K byte loads per lane).
//   * BAND (optional, Smith-Waterman): strip b sweeps only the columns
//     [center(first row) - w, center(last row) + w], center(r) = r * F / R; every cell outside
//     these per-strip rectangles counts as 0.  w >= max(R, F) is the unbanded computation.
//
//   * AFFINE gaps: E lives in registers like H (one value per row, carried along the row); F runs down the
//     column, so the strip's bottom row hands BOTH H and F to the next strip (two boundary rows, two ring pairs).
//
// Traffic per pair: F bytes of reference per strip + 8 bytes per column and strip of boundary
// rows; at 10k x 10k that is 0.007 B per cell -- this path stays VALU bound as well.
#pragma once

#include "dp_kernels.hip.h"
#include "valign_hip.h"

namespace valign {

constexpr int kPhase = 64;                 // steps between ring refills; must be >= G - 1
constexpr int kRing = 2 * kPhase;          // ring slots per lane group
constexpr int kLead = 8;                   // columns the ring refill runs ahead of the phase (the sweep prefetches two)

struct LongArgs {
    const uint8_t *reads;
    const uint8_t *refs;
    int16_t *scores;
    unsigned *brow;            // [2][pair-of-pairs][row_dwords] boundary rows (double buffered by strip)
    long long n;
    long long pp_total;        // pair-of-pairs slots in brow
    int R, F;
    int strips;                // ceil(R / (G*K))
    int row_dwords;            // dwords per boundary row: F rounded up to kPhase, plus kPhase
    int band_half;             // < 0: every column; else strip b only sweeps the columns within band_half of
                               // the diagonal through its rows (cells outside count as 0; SW only)
    short match, mismatch;
    short gap_read, gap_ref;
    short open_read, ext_read, open_ref, ext_ref;     // affine instantiations (Gotoh, as score_kernel's kGapAffine)
};

// DP cell representation of the long-read kernel.  Packed: two pairs per register, int16 (the
// reference's cell type).  Wide: one pair per register, int32 -- for (shape, scoring) whose cells
// could leave int16; the lane group then sweeps its two pairs one after the other.
template <bool WIDE>
struct Cell;
template <>
struct Cell<false> {
    using T = s16x2;
    static __device__ __forceinline__ T bc(int v) { return pk((short)v); }
    static __device__ __forceinline__ T mx(T a, T b) { return pk_max(a, b); }
    static __device__ __forceinline__ T sub0(T a, T g) { return pk_sub_floor0(a, g); }
    static __device__ __forceinline__ T adds(T a, T c) { return pk_add_sat(a, c); }      // "minus infinity" must not wrap
    static __device__ __forceinline__ T ninf() { return pk(kNegInf); }
    static __device__ __forceinline__ unsigned bits(T v) { return as_u32(v); }
    static __device__ __forceinline__ T from_bits(unsigned v) { return as_pk(v); }
};
template <>
struct Cell<true> {
    using T = int;
    static __device__ __forceinline__ T bc(int v) { return v; }
    static __device__ __forceinline__ T mx(T a, T b) { return a > b ? a : b; }
    static __device__ __forceinline__ T sub0(T a, T g) { const int d = a - g; return d > 0 ? d : 0; }
    static __device__ __forceinline__ T adds(T a, T c) { return a + c; }
    static __device__ __forceinline__ T ninf() { return -(1 << 29); }
    static __device__ __forceinline__ unsigned bits(T v) { return (unsigned)v; }
    static __device__ __forceinline__ T from_bits(unsigned v) { return (int)v; }
};

template <int G, int K, bool AFFINE = false, bool SINGLE = false>       // SINGLE: the read is one strip -- no boundary rings
struct LongLds {
    using geo = Geo<G, K>;
    // The ring of reference slab numbers is read across the whole lane skew: lane G - 1 is G - 1 columns behind lane 0,
    // the sweep looks two columns ahead and the refill runs a phase + kLead ahead -- 64 lanes need 256 slots
    static constexpr int kCodeRing = (G - 1) + 2 + kPhase + kLead <= kRing ? kRing : 2 * kRing;
    static constexpr int kCodes = geo::kProfBytes;                        // [groups][kCodeRing][2] bytes
    static constexpr int kIn = kCodes + geo::kGroups * kCodeRing * 2;         // [groups][kRing] dwords
    static constexpr int kOut = kIn + geo::kGroups * kRing * 4;           // [groups][kRing] dwords
    static constexpr int kInF = kOut + geo::kGroups * kRing * 4;          // affine: the F values of the boundary rows, same rings
    static constexpr int kOutF = kInF + geo::kGroups * kRing * 4;
    static constexpr int kTotal = SINGLE ? kIn : (AFFINE ? kOutF + geo::kGroups * kRing * 4 : kInF);      // (LDS bounds the waves per CU: 9 at 16.7 KB, 12 at 12.6)
};

// Columns [c_lo, c_hi] swept by strip s.  c_lo is a multiple of 4 (16-byte ring accesses).
template <int G, int K>
__host__ __device__ inline void strip_columns(int s, int R, int F, int pad_rows, int band_half, int &c_lo, int &c_hi) {
    constexpr int rows = G * K;
    // (a band is only ever swept with the geometry whose strips are the API's blocks: Engine::score_long_device checks
    // G * K == VALIGN_HIP_BAND_BLOCK_ROWS there; taller strips exist for unbanded sweeps)
    static_assert(VALIGN_HIP_BAND_COL_ALIGN == 4, "c_lo is rounded down to a multiple of 4 below");
    if (band_half < 0 || R <= 0) {
        c_lo = 0;
        c_hi = F - 1;
        return;
    }
    int r_lo = s * rows - pad_rows, r_hi = (s + 1) * rows - pad_rows - 1;
    r_lo = r_lo < 0 ? 0 : r_lo;
    r_hi = r_hi > R - 1 ? R - 1 : r_hi;
    const long long lo = (long long)r_lo * F / R - band_half, hi = (long long)r_hi * F / R + band_half;
    c_lo = (int)(lo < 0 ? 0 : lo) & ~3;
    c_hi = (int)(hi > F - 1 ? F - 1 : hi);
}

// AFFINE: Gotoh recurrence (E along the row in registers like H; F down the column -- through the lanes by DPP
// and from strip to strip through a second boundary row next to H's).  SYM then means open_read == open_ref and
// ext_read == ext_ref: H - open is computed once per cell.
// (G == 64: LDS lets eleven one-wave blocks share a CU -- ask the compiler for a register budget that lets three waves share a SIMD)
// F16 (Smith-Waterman, one gap score, int16-packed registers only): the cells are half floats holding value * 2^-10 -- exact
// while every value stays below 1024, which the engine checks (half_float_unit_exact) -- so that "max(x + g, 0)" is ONE
// packed add with the clamp modifier and the cell is a three-operand maximum: add, max3, add-clamp per packed register
// instead of add, max, sub, max, max (score_kernel's kGapSymF16).  What routes here: short reads against a reference too
// long for the resident kernels' LDS.
// SINGLE (the read fits ONE strip: short reads routed here for their reference's length): nothing crosses strips -- no
// boundary rings in LDS (16.7 -> 12.6 KB per wave: 12 waves per CU instead of 9), no ring read / write per step.
template <int G, int K, int ALG, bool SYM, bool WIDE, bool AFFINE = false, bool F16 = false, bool SINGLE = false>
__global__ void __launch_bounds__(64, G == 64 ? 3 : 1)
score_long_kernel(const LongArgs args) {
    static_assert(!F16 || (ALG == kAlgSW && SYM && !WIDE && !AFFINE), "half-float cells: Smith-Waterman, one gap score, packed");
    using geo = Geo<G, K>;
    using lay = LongLds<G, K, AFFINE, SINGLE>;
    using ops = Cell<WIDE>;
    using cell_t = typename ops::T;
    static_assert(kPhase >= G - 1, "a phase must cover the pipeline skew");
    const int lane = threadIdx.x;
    const int grp = lane / G;
    const int l = lane % G;
    const int R = args.R, F = args.F;
    const long long pair0 = (long long)blockIdx.x * geo::kPairs;
    if (pair0 >= args.n) return;                                          // one wave per block
    const int last = (int)((args.n - pair0 < geo::kPairs ? args.n - pair0 : geo::kPairs) - 1);
    const int pad_rows = args.strips * geo::kRows - R;

    unsigned char *prof = valign_smem;
    unsigned char *codes = valign_smem + lay::kCodes;
    unsigned *ring_in = reinterpret_cast<unsigned *>(valign_smem + lay::kIn);
    unsigned *ring_out = reinterpret_cast<unsigned *>(valign_smem + lay::kOut);
    unsigned *ring_in_f = reinterpret_cast<unsigned *>(valign_smem + lay::kInF);
    unsigned *ring_out_f = reinterpret_cast<unsigned *>(valign_smem + lay::kOutF);

    const unsigned lane_base = lds_offset(prof) + l * geo::kLaneBytes;
    const unsigned codes_base = lds_offset(codes) + grp * (lay::kCodeRing * 2);
    const unsigned in_base = lds_offset(ring_in) + grp * (kRing * 4);
    const unsigned in_base_f = lds_offset(ring_in_f) + grp * (kRing * 4);
    unsigned *out_grp = ring_out + grp * kRing;
    unsigned *out_grp_f = ring_out_f + grp * kRing;
    const long long pp0 = pair0 / 2;                                      // first pair-of-pairs of the wave

    const cell_t g_read = ops::bc(ALG == kAlgSW ? -args.gap_read : args.gap_read);
    const cell_t g_ref = ops::bc(ALG == kAlgSW ? -args.gap_ref : args.gap_ref);
    // affine: magnitudes for the SW floor-at-zero subtract, signed addends for the NW variant (as score_kernel)
    const cell_t o_read = ops::bc(ALG == kAlgSW ? -args.open_read : args.open_read), e_read = ops::bc(ALG == kAlgSW ? -args.ext_read : args.ext_read);
    const cell_t o_ref = ops::bc(ALG == kAlgSW ? -args.open_ref : args.open_ref), e_ref = ops::bc(ALG == kAlgSW ? -args.ext_ref : args.ext_ref);
    const cell_t border_f = (AFFINE && ALG == kAlgNW) ? ops::ninf() : ops::bc(0);     // gap matrices start at minus infinity (NW)
    const unsigned border_f_bits = ops::bits(border_f);
    // boundary-row slot of lane group g: H, and next to it F (affine); wide cells keep one per half
    auto brow_slot_of = [&](int g, int half_, int is_f) __attribute__((always_inline)) -> long long {
        return ((pp0 + g) * (WIDE ? 2 : 1) + (WIDE ? half_ : 0)) * (AFFINE ? 2 : 1) + is_f;
    };

  // wide cells: the group's two pairs take turns (half 0, then half 1); packed cells: one pass
  for (int half = 0; half < (WIDE ? 2 : 1); ++half) {
    cell_t best = ops::bc(0), col_best = ops::bc(0), row_best = ops::bc(0);
    const long long brow_slot = brow_slot_of(grp, half, 0);                    // boundary row of this sweep

    for (int i = lane; i < geo::kPairStride / 4; i += kWave)
        reinterpret_cast<unsigned *>(prof + geo::kZeroSlab * geo::kPairStride)[i] = 0u;
    // the sweep prefetches profile rows through whatever slab numbers the ring holds: start with valid ones
    for (int i = lane; i < geo::kGroups * lay::kCodeRing * 2; i += kWave) codes[i] = (unsigned char)geo::kZeroSlab;

    for (int s = 0; s < args.strips; ++s) {
        // ---- query profile of this strip's rows ----
        __syncthreads();                       // the previous strip is done with the profile
#pragma unroll
        for (int i = 0; i < 2 * K; ++i) {
            const int idx = lane + kWave * i;
            const int p = idx / geo::kRows, rr = idx - p * geo::kRows;
            const int ps = p > last ? last : p;
            const int grow = s * geo::kRows + rr - pad_rows;              // read position of this row
            const int a = (grow >= 0 && grow < R) ? base_class(args.reads[(pair0 + ps) * R + grow]) : 0;
            const bool valid = a >= 1 && a <= 4;
            const int off = p * geo::kPairStride + geo::row_offset(rr / K, rr % K);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                short sc = valid ? (a == c + 1 ? args.match : args.mismatch) : (short)0;
                if (F16) sc = __builtin_bit_cast(short, (_Float16)((float)sc * (1.0f / 1024.0f)));
                *reinterpret_cast<short *>(prof + c * geo::kPairStride * geo::kPairs + off) = sc;
            }
        }
        // columns swept by this strip [c_lo, c_hi] and by the previous one [p_lo, p_hi]
        int c_lo, c_hi, p_lo = 0, p_hi = -1;
        strip_columns<G, K>(s, R, F, pad_rows, args.band_half, c_lo, c_hi);
        if (s > 0) strip_columns<G, K>(s - 1, R, F, pad_rows, args.band_half, p_lo, p_hi);
        const int ncols = c_hi - c_lo + 1;
        const int steps = ncols + G - 1;
        const unsigned *brow_prev = args.brow + (long long)((s & 1) ^ 1) * args.pp_total * args.row_dwords;
        unsigned *brow_cur = args.brow + (long long)(s & 1) * args.pp_total * args.row_dwords;

        cell_t Hl[K];
        int Gl[(SYM && WIDE && ALG == kAlgSW) ? K : 1];       // int32 SW: max(h - g, 0) of the previous column
#pragma unroll
        for (int q = 0; q < K; ++q) Hl[q] = ops::bc(0);
#pragma unroll
        for (int q = 0; q < ((SYM && WIDE && ALG == kAlgSW) ? K : 1); ++q) Gl[q] = 0;
        cell_t El[(AFFINE || F16) ? K : 1];     // affine: E of the previous column; F16: max(h + g, 0) of it
#pragma unroll
        for (int q = 0; q < ((AFFINE || F16) ? K : 1); ++q) El[q] = border_f;
        // AFFINE && SYM (same open / extend both ways): H - open of the previous column, computed once per cell and
        // shared by E of this column and F of the next row (score_kernel's kGapAffineSym)
        cell_t HOl[(AFFINE && SYM) ? K : 1];
#pragma unroll
        for (int q = 0; q < ((AFFINE && SYM) ? K : 1); ++q)
            HOl[q] = (ALG == kAlgSW) ? ops::sub0(ops::bc(0), o_ref) : ops::adds(ops::bc(0), o_ref);
        cell_t f_last = border_f;
        cell_t up0 = ops::bc(0), h_last = ops::bc(0);
        if (l == 0 && c_lo - 1 >= p_lo && c_lo - 1 <= p_hi)     // diagonal neighbour of the first swept column
            up0 = ops::from_bits(__builtin_nontemporal_load(brow_prev + brow_slot * args.row_dwords + (c_lo - 1)));
        int j = c_lo - l;
        unsigned pa[K / 2], pb[K / 2];            // raw profile dwords of the coming step (LDS pipeline)
        unsigned ca_next = 0, cb_next = 0;        // slab numbers of the step after

        // TRACK: how the step feeds the running SW maximum (see score_kernel): every diag + S, or, for
        // the shared-gap recurrence in the unmasked step pairs, nothing in the first step and every
        // max(left, up) plus the last row in the second.
        auto step = [&](auto masked_tag, auto track_tag, int t) __attribute__((always_inline)) {
            constexpr bool MASKED = decltype(masked_tag)::value;
            constexpr int TRACK = decltype(track_tag)::value;
            const cell_t diag0 = up0;
            // row above: previous lane of the group; for the first lane the previous strip's bottom row
            // (one instruction: the shift writes every lane that has a lane before it -- G = 64: all but lane 0, which keeps
            // the "old" operand, the ring's value; narrower groups select)
            const unsigned from_ring = SINGLE ? 0u : *(lds_cu32 *)(in_base + (((c_lo + t) & (kRing - 1)) << 2));
            if constexpr (SINGLE && G == 16) {              // (row_shr:1: the group's first lane reads 0, the border)
                up0 = ops::from_bits((unsigned)__builtin_amdgcn_update_dpp(0, (int)ops::bits(h_last), 0x111, 0xF, 0xF, true));
            } else if constexpr (G == kWave) {
                up0 = ops::from_bits((unsigned)__builtin_amdgcn_update_dpp((int)from_ring, (int)ops::bits(h_last), 0x138, 0xF, 0xF, false));
            } else {
                const unsigned from_lane = (unsigned)__builtin_amdgcn_update_dpp(0, (int)ops::bits(h_last), 0x138, 0xF, 0xF, true);
                up0 = ops::from_bits(l == 0 ? from_ring : from_lane);
            }
            cell_t fup0 = border_f;
            if constexpr (AFFINE) {
                const unsigned f_ring = SINGLE ? border_f_bits : *(lds_cu32 *)(in_base_f + (((c_lo + t) & (kRing - 1)) << 2));
                if constexpr (G == kWave) {
                    fup0 = ops::from_bits((unsigned)__builtin_amdgcn_update_dpp((int)f_ring, (int)ops::bits(f_last), 0x138, 0xF, 0xF, false));
                } else {
                    const unsigned f_lane = (unsigned)__builtin_amdgcn_update_dpp(0, (int)ops::bits(f_last), 0x138, 0xF, 0xF, true);
                    fup0 = ops::from_bits(l == 0 ? f_ring : f_lane);
                }
            }
            // LDS fetches run ahead of the arithmetic (every lane, every step): the raw profile dwords of
            // this step are in registers, the rows of step t+1 and the slab numbers of step t+2 are
            // requested now.  The ring refill leads the phase by kLead columns for that.
            cell_t S[K];
            if constexpr (WIDE) {
#pragma unroll
                for (int c = 0; c < K / 2; ++c) {
                    S[2 * c] = (int)(short)(pa[c] & 0xFFFFu);
                    S[2 * c + 1] = (int)pa[c] >> 16;
                }
                lds_load_lane<K>(lane_base + (half ? cb_next : ca_next) * geo::kPairStride, pa);
            } else {
                merge_profile<K>(pa, pb, S);
                lds_load_lane<K>(lane_base + ca_next * geo::kPairStride, pa);
                lds_load_lane<K>(lane_base + cb_next * geo::kPairStride, pb);
            }
            {
                const unsigned next_addr = codes_base + (((j + 2) & (lay::kCodeRing - 1)) << 1);
                ca_next = *(lds_cu8 *)(next_addr);
                cb_next = *(lds_cu8 *)(next_addr + 1);
            }
            if (!MASKED || (unsigned)(j - c_lo) < (unsigned)ncols) {
                // column-independent work of row q+1 sits between the links of the dependent chain of
                // row q (see score_kernel)
                cell_t h = up0;
                if constexpr (AFFINE) {
                    // E, diag + S and their maximum of row q + 1 only need the previous column: computed one row
                    // ahead, between the links of the dependent chain F -> H down the column (see score_kernel)
                    auto pass1 = [&](int q) __attribute__((always_inline)) -> cell_t {
                        const cell_t d = (q == 0 ? diag0 : Hl[q - 1]) + S[q];
                        cell_t e;
                        if constexpr (SYM)
                            e = ops::mx((ALG == kAlgSW) ? ops::sub0(El[q], e_read) : ops::adds(El[q], e_read), HOl[q]);
                        else
                            e = (ALG == kAlgSW) ? ops::mx(ops::sub0(El[q], e_read), ops::sub0(Hl[q], o_read))
                                                : ops::mx(ops::adds(El[q], e_read), ops::adds(Hl[q], o_read));
                        El[q] = e;
                        if (ALG == kAlgSW) best = ops::mx(best, d);
                        return ops::mx(d, e);
                    };
                    cell_t f = fup0;
                    cell_t ho = (ALG == kAlgSW) ? ops::sub0(up0, o_ref) : ops::adds(up0, o_ref);      // SYM: H - open of the row above
                    cell_t m_cur = pass1(0);
#pragma unroll
                    for (int q = 0; q < K; ++q) {
                        if constexpr (SYM)
                            f = ops::mx((ALG == kAlgSW) ? ops::sub0(f, e_ref) : ops::adds(f, e_ref), ho);
                        else
                            f = (ALG == kAlgSW) ? ops::mx(ops::sub0(f, e_ref), ops::sub0(h, o_ref))
                                                : ops::mx(ops::adds(f, e_ref), ops::adds(h, o_ref));
                        cell_t m_next = ops::bc(0);
                        if (q + 1 < K) m_next = pass1(q + 1);          // before Hl[q] (and HOl[q]) are overwritten
                        h = ops::mx(m_cur, f);
                        Hl[q] = h;
                        if constexpr (SYM) {
                            ho = (ALG == kAlgSW) ? ops::sub0(h, o_ref) : ops::adds(h, o_ref);
                            HOl[q] = ho;
                        }
                        m_cur = m_next;
                    }
                    f_last = f;
                } else if constexpr (F16) {
                    auto hf = [](s16x2 v) __attribute__((always_inline)) { return __builtin_bit_cast(f16x2, v); };
                    auto hb = [](f16x2 v) __attribute__((always_inline)) { return __builtin_bit_cast(s16x2, v); };
                    const _Float16 gs = (_Float16)((float)args.gap_ref * (1.0f / 1024.0f));
                    const f16x2 g_unit = f16x2{gs, gs}, zero2 = f16x2{(_Float16)0, (_Float16)0}, one2 = f16x2{(_Float16)1, (_Float16)1};
                    // (the row above arrives as h alone -- from the lane before or from the previous strip's boundary row)
                    f16x2 up_c = __builtin_elementwise_min(__builtin_elementwise_max(hf(up0) + g_unit, zero2), one2);
                    f16x2 hh = zero2;
                    f16x2 d_cur = hf(diag0) + hf(S[0]), d_prev = zero2;
                    f16x2 bestf = hf(best);
#pragma unroll
                    for (int q = 0; q < K; ++q) {
                        f16x2 d_next = d_cur;
                        if (q + 1 < K) d_next = hf(Hl[q]) + hf(S[q + 1]);       // before Hl[q] is overwritten
                        hh = __builtin_elementwise_maximum(__builtin_elementwise_maximum(d_cur, hf(El[q])), up_c);
                        Hl[q] = hb(hh);
                        up_c = __builtin_elementwise_min(__builtin_elementwise_max(hh + g_unit, zero2), one2);   // v_pk_add_f16 clamp
                        El[q] = hb(up_c);
                        if (q & 1) bestf = __builtin_elementwise_maximum(__builtin_elementwise_maximum(bestf, d_prev), d_cur);
                        else if (q == K - 1) bestf = __builtin_elementwise_maximum(bestf, d_cur);
                        d_prev = d_cur;
                        d_cur = d_next;
                    }
                    best = hb(bestf);
                    h = hb(hh);
                } else if constexpr (SYM && WIDE && ALG == kAlgSW) {
                    // int32 cells have a three-operand maximum and a saturating subtract: each cell keeps
                    // (h, max(h - g, 0)) and h = max3(diag + S, left', up') on the floored registers is
                    // non-negative by construction -- add, max3, sub-clamp per cell instead of add, max, sub,
                    // max, max (the packed int16 form has no max3).  The maximum takes two rows per max3.
                    const unsigned gmag = (unsigned)g_ref;
                    int up_c = (int)__builtin_elementwise_sub_sat((unsigned)up0, gmag);
                    int d_cur = diag0 + S[0], d_prev = 0;
#pragma unroll
                    for (int q = 0; q < K; ++q) {
                        int d_next = 0;
                        if (q + 1 < K) d_next = Hl[q] + S[q + 1];          // before Hl[q] is overwritten
                        int m = d_cur > Gl[q] ? d_cur : Gl[q];
                        m = m > up_c ? m : up_c;
                        h = m;
                        Hl[q] = m;
                        up_c = (int)__builtin_elementwise_sub_sat((unsigned)m, gmag);
                        Gl[q] = up_c;
                        if (q & 1) {
                            int b2 = best > d_prev ? best : d_prev;
                            best = b2 > d_cur ? b2 : d_cur;
                        } else if (q == K - 1) {
                            best = best > d_cur ? best : d_cur;
                        }
                        d_prev = d_cur;
                        d_cur = d_next;
                    }
                } else if (SYM) {
                    cell_t d_cur = diag0 + S[0];
#pragma unroll
                    for (int q = 0; q < K; ++q) {
                        const cell_t x = ops::mx(Hl[q], h);
                        cell_t d_next = ops::bc(0);
                        if (q + 1 < K) d_next = Hl[q] + S[q + 1];
                        const cell_t y = (ALG == kAlgSW) ? ops::sub0(x, g_ref) : x + g_ref;
                        if (ALG == kAlgSW && TRACK == kTrackAll) best = ops::mx(best, d_cur);
                        if (ALG == kAlgSW && TRACK == kTrackPair) best = ops::mx(best, x);
                        h = ops::mx(d_cur, y);
                        Hl[q] = h;
                        d_cur = d_next;
                        __builtin_amdgcn_sched_barrier(0);      // keep the interleaving: the scheduler regroups it otherwise
                    }
                    if (ALG == kAlgSW && TRACK == kTrackPair) best = ops::mx(best, h);
                } else {
                    auto pass1 = [&](int q) __attribute__((always_inline)) -> cell_t {
                        const cell_t d = (q == 0 ? diag0 : Hl[q - 1]) + S[q];
                        const cell_t e = (ALG == kAlgSW) ? ops::sub0(Hl[q], g_read) : Hl[q] + g_read;
                        if (ALG == kAlgSW) best = ops::mx(best, d);
                        return ops::mx(d, e);
                    };
                    cell_t m_cur = pass1(0);
#pragma unroll
                    for (int q = 0; q < K; ++q) {
                        const cell_t f = (ALG == kAlgSW) ? ops::sub0(h, g_ref) : h + g_ref;
                        cell_t m_next = ops::bc(0);
                        if (q + 1 < K) m_next = pass1(q + 1);
                        h = ops::mx(m_cur, f);
                        Hl[q] = h;
                        m_cur = m_next;
                    }
                }
                h_last = h;
                if (l == G - 1) {
                    if constexpr (!SINGLE) {
                        out_grp[j & (kRing - 1)] = ops::bits(h);           // bottom row of the strip
                        if constexpr (AFFINE) out_grp_f[j & (kRing - 1)] = ops::bits(f_last);
                    }
                    if (ALG == kAlgNW) row_best = ops::mx(row_best, h);
                }
            }
            ++j;
        };

        // ---- ring refill, software pipelined ----
        // The reference bases and the previous strip's boundary values of phase p+1 are requested
        // from HBM/L2 when phase p starts computing (prefetch: global -> registers) and reach the
        // LDS rings when it is done (commit): a wave never waits for its own refill, which cost a
        // third of the cycles when the loads were issued and consumed in one go.
        unsigned char pre_base[8];
        u32x4 pre_brow = {0u, 0u, 0u, 0u}, pre_frow = {0u, 0u, 0u, 0u};
        bool pre_brow_valid = false;
        auto prefetch = [&](int t0) __attribute__((always_inline)) {
            // (lanes beyond the wave's pairs / groups have nothing to fetch: G = 64 has one group of two pairs)
            const int p = lane / 8, c0 = c_lo + kLead + t0 + (lane % 8) * 8;  // lane -> pair lane/8, eight columns
            const int ps = p > last ? last : p;
            const uint8_t *src = args.refs + (pair0 + ps) * F;
            if (p < geo::kPairs) {
#pragma unroll
                for (int x = 0; x < 8; ++x) pre_base[x] = (c0 + x < F) ? src[c0 + x] : (unsigned char)0;
            }
            if constexpr (SINGLE) return;
            const int g = lane / 16, col = c_lo + kLead + t0 + (lane % 16) * 4;   // lane -> group lane/16, four columns
            pre_brow_valid = g < geo::kGroups && s > 0 && col + 4 <= args.row_dwords;
            if (pre_brow_valid) {       // L2-served load: the same addresses were read two strips ago and rewritten since
                pre_brow = __builtin_nontemporal_load(
                    reinterpret_cast<const u32x4 *>(brow_prev + brow_slot_of(g, half, 0) * args.row_dwords + col));
                if constexpr (AFFINE)
                    pre_frow = __builtin_nontemporal_load(
                        reinterpret_cast<const u32x4 *>(brow_prev + brow_slot_of(g, half, 1) * args.row_dwords + col));
            }
        };
        auto commit = [&](int t0) __attribute__((always_inline)) {
            {
                const int p = lane / 8, c0 = c_lo + kLead + t0 + (lane % 8) * 8;
                unsigned char *dst = codes + (p / 2) * (lay::kCodeRing * 2) + (p & 1);
                if (p < geo::kPairs) {
#pragma unroll
                    for (int x = 0; x < 8; ++x) {
                        const int col = c0 + x;
                        const int c = col < F ? base_class(pre_base[x]) : 0;
                        dst[(col & (lay::kCodeRing - 1)) * 2] =
                            (unsigned char)((c >= 1 && c <= 4) ? (c - 1) * geo::kPairs + p : geo::kZeroSlab);
                    }
                }
                if constexpr (SINGLE) return;
                const int g = lane / 16, col = c_lo + kLead + t0 + (lane % 16) * 4;
                if (g >= geo::kGroups) return;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (pre_brow_valid) {
                    v.x = (col + 0 >= p_lo && col + 0 <= p_hi) ? pre_brow.x : 0u;      // outside the previous strip's
                    v.y = (col + 1 >= p_lo && col + 1 <= p_hi) ? pre_brow.y : 0u;      // columns the row above is 0
                    v.z = (col + 2 >= p_lo && col + 2 <= p_hi) ? pre_brow.z : 0u;
                    v.w = (col + 3 >= p_lo && col + 3 <= p_hi) ? pre_brow.w : 0u;
                }
                *reinterpret_cast<uint4 *>(ring_in + g * kRing + (col & (kRing - 1))) = v;
                if constexpr (AFFINE) {                                  // F of the row above: the border value where there is none
                    uint4 vf = make_uint4(border_f_bits, border_f_bits, border_f_bits, border_f_bits);
                    if (pre_brow_valid) {
                        vf.x = (col + 0 >= p_lo && col + 0 <= p_hi) ? pre_frow.x : border_f_bits;
                        vf.y = (col + 1 >= p_lo && col + 1 <= p_hi) ? pre_frow.y : border_f_bits;
                        vf.z = (col + 2 >= p_lo && col + 2 <= p_hi) ? pre_frow.z : border_f_bits;
                        vf.w = (col + 3 >= p_lo && col + 3 <= p_hi) ? pre_frow.w : border_f_bits;
                    }
                    *reinterpret_cast<uint4 *>(ring_in_f + g * kRing + (col & (kRing - 1))) = vf;
                }
                if (t0 >= 2 * kPhase && s + 1 < args.strips) {           // drain what lane G-1 finished two phases ago
                    const int oc = col - kLead - 2 * kPhase;
                    *reinterpret_cast<uint4 *>(brow_cur + brow_slot_of(g, half, 0) * args.row_dwords + oc) =
                        *reinterpret_cast<const uint4 *>(ring_out + g * kRing + (oc & (kRing - 1)));
                    if constexpr (AFFINE)
                        *reinterpret_cast<uint4 *>(brow_cur + brow_slot_of(g, half, 1) * args.row_dwords + oc) =
                            *reinterpret_cast<const uint4 *>(ring_out_f + g * kRing + (oc & (kRing - 1)));
                }
            }
        };
        // head of the strip: the first kLead columns go in directly (one column per lane), then the
        // sweep's own pipeline is primed with the slab numbers / profile rows of steps 0 and 1
        {
            const int p = lane / 8, col = c_lo + (lane % 8);
            const int ps = p > last ? last : p;
            if (p < geo::kPairs) {
                const int c = col < F ? base_class(args.refs[(pair0 + ps) * F + col]) : 0;
                codes[(p / 2) * (lay::kCodeRing * 2) + (p & 1) + (col & (lay::kCodeRing - 1)) * 2] =
                    (unsigned char)((c >= 1 && c <= 4) ? (c - 1) * geo::kPairs + p : geo::kZeroSlab);
            }
            if (!SINGLE && lane < geo::kGroups * kLead) {
                const int g = lane / kLead, bc = c_lo + (lane % kLead);
                unsigned v = 0u, vf = border_f_bits;
                if (s > 0 && bc < args.row_dwords && bc >= p_lo && bc <= p_hi) {
                    v = __builtin_nontemporal_load(brow_prev + brow_slot_of(g, half, 0) * args.row_dwords + bc);
                    if constexpr (AFFINE) vf = __builtin_nontemporal_load(brow_prev + brow_slot_of(g, half, 1) * args.row_dwords + bc);
                }
                ring_in[g * kRing + (bc & (kRing - 1))] = v;
                if constexpr (AFFINE) ring_in_f[g * kRing + (bc & (kRing - 1))] = vf;
            }
        }
        prefetch(0);
        for (int t0 = 0; t0 < steps; t0 += kPhase) {
            commit(t0);
            if (t0 == 0) {
                __syncthreads();
                const unsigned a0 = codes_base + ((j & (lay::kCodeRing - 1)) << 1), a1 = codes_base + (((j + 1) & (lay::kCodeRing - 1)) << 1);
                const unsigned ca = *(lds_cu8 *)(a0), cb = *(lds_cu8 *)(a0 + 1);
                if constexpr (WIDE) {
                    lds_load_lane<K>(lane_base + (half ? cb : ca) * geo::kPairStride, pa);
                } else {
                    lds_load_lane<K>(lane_base + ca * geo::kPairStride, pa);
                    lds_load_lane<K>(lane_base + cb * geo::kPairStride, pb);
                }
                ca_next = *(lds_cu8 *)(a1);
                cb_next = *(lds_cu8 *)(a1 + 1);
            }
            __syncthreads();
            if (t0 + kPhase < steps) prefetch(t0 + kPhase);
            const int t1 = t0 + kPhase < steps ? t0 + kPhase : steps;
            using all_t = std::integral_constant<int, kTrackAll>;
            using first_t = std::integral_constant<int, (SYM && ALG == kAlgSW) ? kTrackNone : kTrackAll>;
            using second_t = std::integral_constant<int, (SYM && ALG == kAlgSW) ? kTrackPair : kTrackAll>;
            if (t0 >= G - 1 && t1 <= ncols) {
                int t = t0;
                for (; t + 1 < t1; t += 2) {        // two steps per trip (loop-carried registers swap roles)
                    step(std::false_type{}, first_t{}, t);
                    step(std::false_type{}, second_t{}, t + 1);
                }
                for (; t < t1; ++t) step(std::false_type{}, all_t{}, t);
            } else {
                for (int t = t0; t < t1; ++t) step(std::true_type{}, all_t{}, t);
            }
        }
        // ---- drain the last two phases of the outgoing row ----
        __syncthreads();
        if (!SINGLE && s + 1 < args.strips) {
            const int phases = (steps + kPhase - 1) / kPhase;
            const int g = lane / 16;
            for (int ph = phases - 2 < 0 ? 0 : phases - 2; ph < phases; ++ph) {
                const int oc = c_lo + ph * kPhase + (lane % 16) * 4;
                if (g < geo::kGroups && oc + 4 <= args.row_dwords) {
                    *reinterpret_cast<uint4 *>(brow_cur + brow_slot_of(g, half, 0) * args.row_dwords + oc) =
                        *reinterpret_cast<const uint4 *>(ring_out + g * kRing + (oc & (kRing - 1)));
                    if constexpr (AFFINE)
                        *reinterpret_cast<uint4 *>(brow_cur + brow_slot_of(g, half, 1) * args.row_dwords + oc) =
                            *reinterpret_cast<const uint4 *>(ring_out_f + g * kRing + (oc & (kRing - 1)));
                }
            }
        }
        if (ALG == kAlgNW) {                    // every lane froze at the last column: this strip's rows
#pragma unroll
            for (int q = 0; q < K; ++q) col_best = ops::mx(col_best, Hl[q]);
            if (s + 1 < args.strips) row_best = ops::bc(0);  // only the last strip holds the last row
        }
    }

    cell_t res;
    if constexpr (F16) {
        const f16x2 b = __builtin_bit_cast(f16x2, best);
        res = s16x2{(short)(int)((float)b.x * 1024.0f), (short)(int)((float)b.y * 1024.0f)};
    } else if (ALG == kAlgSW) {
        res = best;
    } else {
        res = ops::mx(col_best, l == G - 1 ? row_best : ops::bc(0));
        res = ops::mx(res, ops::bc(0));
    }
#pragma unroll
    for (int dd = G / 2; dd >= 1; dd >>= 1)
        res = ops::mx(res, ops::from_bits((unsigned)__shfl_xor((int)ops::bits(res), dd, kWave)));
    if (l == 0) {
        const long long pa = pair0 + 2 * grp;
        if constexpr (WIDE) {             // the ABI's score is a short: saturate what it cannot carry
            const int v = res > 32767 ? 32767 : res;
            if (pa + half < args.n) args.scores[pa + half] = (int16_t)v;
        } else {
            if (pa < args.n) args.scores[pa] = res.x;
            if (pa + 1 < args.n) args.scores[pa + 1] = res.y;
        }
    }
  }
}

}  // namespace valign
