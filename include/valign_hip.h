/*
 * valign_hip.h -- entry points of libHIPKernel.so, the MI355X (gfx950) backend.
 *
 * (1) The versalignLib plugin boundary -- what the reference host binds with dlsym()
 *     (src/util/versalignUtil.cpp:35-76, src/impl/main.cpp:217-238).  Same four symbols
 *     and the same behaviour as src/Kernels/default/DefaultKernel_dllexport.cpp:18-42:
 *
 *       AlignmentKernel *spawn_alignment_kernel();
 *       void set_parameters(AlignmentParameters *);
 *       void set_logger(AlignmentLogger *);
 *       void delete_alignment_kernel(AlignmentKernel *);
 *
 *     The spawned object implements AlignmentKernel::score_alignments and
 *     ::compute_alignments (include/AlignmentKernel.h:40-43, restated in
 *     versalign_plugin_abi.h).  Required parameter keys are the reference's six
 *     (DefaultKernel.h:70-75); a missing one makes spawn throw the same C string.
 *     Optional keys, probed with has_key (a reference host simply lacks them):
 *       score_gap_open_read / score_gap_extend_read /
 *       score_gap_open_ref  / score_gap_extend_ref ... affine-gap extension (scores and alignments)
 *       traceback_policy ................................ 0 Default/OpenCL tie-breaks (default),
 *                                                         1 SSE2/AVX2 tie-breaks (linear gaps)
 *       band_width ...................................... > 0: banded Smith-Waterman scores, that
 *                                                         many diagonals around the main one (strip band)
 *       score_width ..................................... DP cells of score_alignments: 0 auto (int16,
 *                                                         int32 where int16 could overflow), 16, 32
 *       ragged_batching ................................. length-sorted score calls, both modes
 *                                                         (trailing non-ACGT padding is not swept,
 *                                                         identical scores): 0 never (default), 1 when a
 *                                                         sample of the call is ragged enough, 2 always
 *       host_malloc_tuning .............................. 0 (default): the plugin leaves the HOST's allocator alone.
 *                                                         A host that wants compute_alignments' 2n operator new[]
 *                                                         result rows cheap (~50 ms instead of ~700 ms per million
 *                                                         pairs with 16 threads) starts with MALLOC_TOP_PAD_=268435456
 *                                                         in its environment -- or sets this key:
 *                                                         2 = mallopt(M_TOP_PAD, 256 MB) from the plugin (process-wide,
 *                                                         logged at WARNING level the first time), 1 = that +
 *                                                         M_TRIM_THRESHOLD off.  Other values are refused.
 *       host_packing .................................... score_alignments: 1 (default) sequences cross PCIe as 4-bit
 *                                                         base classes (identical scores), 0 raw ASCII
 *       half_float_cells ................................ score_alignments: 1 (default) half-float cells where they are
 *                                                         exact (identical scores, fewer instructions), 0 integer cells
 *       pointer_scratch_cap_mb .......................... cap of compute_alignments' device-side pointer
 *                                                         scratch in MiB (default 0: 64 GiB / half the free HBM)
 *       hip_device ...................................... device ordinal (default 0)
 *       hip_devices ..................................... N > 1: every call is split into N contiguous shards of
 *                                                         pairs, one per device (hip_device .. hip_device+N-1,
 *                                                         modulo the visible ones -- folding is announced at
 *                                                         WARNING level), each on its own host thread;
 *                                                         results land in the caller's arrays (default 1)
 *       hip_devices_strict .............................. 1: refuse hip_devices beyond the visible devices
 *       hip_devices_allgather ........................... 1: score_alignments keeps every shard's scores on its device and
 *                                                         runs an RCCL all-gather over them (librccl.so, loaded with dlopen;
 *                                                         one communicator per device in this process): the whole score
 *                                                         vector ends up on every device, the host copy comes from the
 *                                                         first.  Needs one distinct device per shard.  Default 0: every
 *                                                         shard copies its own scores to the host, no collective
 *       hip_group_lanes / hip_rows_per_lane ............. force a kernel geometry
 *
 * (2) A flat C view of the same engine for callers that already hold the batch in
 *     device memory (bench.py, multi-GPU sharding): plain pointers and sizes only.
 *     These replace nothing in the reference -- its OpenCL backend is the closest
 *     precedent (gather -> contiguous pair-major buffers -> device,
 *     src/Kernels/OpenCL/OpenCLKernel.cpp:57-108) -- and take exactly the contiguous
 *     buffers that backend builds internally.
 *
 * All flat functions return 0 on success, non-zero on failure (message through
 * valign_hip_last_error()).  There is no CPU fallback anywhere in this library.
 */
#ifndef VALIGN_HIP_H
#define VALIGN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- (1) plugin boundary (C++ types from versalign_plugin_abi.h) ---- */
#ifdef __cplusplus
class AlignmentKernel;
class AlignmentParameters;
class AlignmentLogger;
AlignmentKernel *spawn_alignment_kernel();
void set_parameters(AlignmentParameters *parameters);
void set_logger(AlignmentLogger *logger);
void delete_alignment_kernel(AlignmentKernel *instance);
#endif

/* ---- (2) flat device-resident API ---- */
typedef struct valign_hip_engine valign_hip_engine;

typedef struct {
    int32_t match, mismatch;          /* score_match, score_mismatch                      */
    int32_t gap_read, gap_ref;        /* score_gap_read (LEFT), score_gap_ref (UP)        */
    int32_t affine;                   /* 0: linear model above; 1: use the four below     */
    int32_t open_read, ext_read;      /* first / further gap base in the read (LEFT)      */
    int32_t open_ref, ext_ref;        /* first / further gap base in the ref  (UP)        */
} valign_hip_scoring;

int valign_hip_device_count(void);

/* Plugin key hip_devices = N: shard d (0-based) of a call of n pairs is the contiguous range [*begin, *begin + *count)
 * -- ceil(n / N) pairs per shard, the last one short, trailing shards empty.  Pure arithmetic (no device needed): the one
 * rule the shard threads and the in-plugin all-gather's buffer offsets both use.  Returns 1 on bad arguments.            */
int valign_hip_shard_range(int n, int shards, int d, int *begin, int *count);

/* One engine = one device + fixed (read_length, ref_length, scoring), like one spawned
 * kernel object.  force_group_lanes / force_rows_per_lane = 0 lets the engine choose. */
int valign_hip_engine_create(int device, int read_length, int ref_length,
                             const valign_hip_scoring *scoring, int force_group_lanes,
                             int force_rows_per_lane, valign_hip_engine **out);
void valign_hip_engine_destroy(valign_hip_engine *e);

/* Tie-break rules of compute_alignments: 0 = the reference's Default/OpenCL kernels (default),
 * 1 = its SSE2/AVX2 kernels (DIAG only between ACGT bases > LEFT > UP, no stop at zero cells,
 * N invalid for the NW end cell; src/Kernels/AVX-SSE/SSEKernel.cpp:366-379, 532-536).          */
int valign_hip_set_traceback_policy(valign_hip_engine *e, int policy);

/* Banded Smith-Waterman scores (an extension: the reference has no banding).  Definition, w = diagonals / 2:
 * the per-cell band is |j - floor(i * ref_length / read_length)| <= w (i, j 0-based read / ref positions).
 * The library computes AT LEAST that band: the read's rows are taken in blocks of B consecutive rows (the last
 * block ends with the last row), and a block computes the columns [floor(r_first * F / R) - w,
 * floor(r_last * F / R) + w] of its rows r_first..r_last, the lower end rounded down to a multiple of A; every
 * other cell counts as 0 and cannot hold the maximum.  Hence  score(per-cell band w) <= result <= score(all
 * cells),  and the result equals this block definition exactly (oracle/cpu_ref.c, vref_score_banded_sw[_affine],
 * restates it with B and A as parameters; B = 1 / A = 1 is the per-cell band).
 *
 * WHICH (B, A) applies depends on the schedule the engine can use for (shape, band, scoring), and
 * valign_hip_describe is the authority -- "band_block_rows" / "band_col_align" of the engine after
 * valign_hip_set_band_width:
 *   B = 16,  A = 1   the cyclic block chain (band_kernels.hip.h): linear gaps since round 3, affine gaps since round 4,
 *                    wherever its plan fits (windows up to 2048 columns of reference ring, delays up to 64 steps);
 *                    exported below as VALIGN_HIP_BAND_CHAIN_BLOCK_ROWS / VALIGN_HIP_BAND_CHAIN_COL_ALIGN;
 *   B = 160, A = 4   row strips (long_kernels.hip.h): every other banded case; VALIGN_HIP_BAND_BLOCK_ROWS /
 *                    VALIGN_HIP_BAND_COL_ALIGN.
 * The chain's band is the tighter superset of the per-cell band.  0 (default) computes every cell; a band wider
 * than the matrix gives the unbanded result.                                                                   */
#define VALIGN_HIP_BAND_BLOCK_ROWS 160
#define VALIGN_HIP_BAND_COL_ALIGN 4
#define VALIGN_HIP_BAND_CHAIN_BLOCK_ROWS 16
#define VALIGN_HIP_BAND_CHAIN_COL_ALIGN 1
int valign_hip_set_band_width(valign_hip_engine *e, int diagonals);

/* Cap (MiB) of the internal pointer scratch compute_alignments keeps in device memory (2 bits per cell and pair,
 * 4 with affine gaps: 20.8 / 41.6 KB per pair at 150 x 500).  0 (default): up to 64 GiB or half the free HBM,
 * whichever is smaller; batches that need more than the cap run in chunks -- same results, more launches.
 * Plugin key: pointer_scratch_cap_mb.                                                                       */
int valign_hip_set_pointer_scratch_cap_mb(valign_hip_engine *e, long long mb);

/* DP cell width of the score path: 0 (default) = int16 like the reference, switching to int32 cells
 * for (shape, scoring, mode) whose cells could leave int16 (the reference would wrap silently);
 * 16 = int16 or refuse; 32 = always int32 (strip path, one pair per register: half the rate).
 * Scores beyond the ABI's short saturate at 32767.                                               */
int valign_hip_set_score_width(valign_hip_engine *e, int bits);

/* Length-sorted batching of score calls (Smith-Waterman and the NW variant; valign_hip_score_host / score_alignments
 * and valign_hip_score_device): the reference host pads every sequence to the longest
 * (src/util/versalignUtil.cpp:17-33) and every backend sweeps the padding; here pairs are binned by their length
 * without trailing non-ACGT bytes and each bin is swept at its own shape.  Scores are identical: trailing padding
 * scores 0, so it cannot raise a Smith-Waterman maximum, and every value of the real matrix's last row / column runs
 * down its diagonal unchanged to the padded matrix's, which is where the NW variant reads its result.
 * Classification, packing by length class and the scores' way back run on the device.  0 = never (default), 1 = when the
 * call is ragged enough to skip a third of the cells (host pointers: judged from a sample of the call's tails;
 * valign_hip_score_device: from the device's histogram), 2 = always.  With mode 1 or 2 valign_hip_score_device WAITS
 * for the classification of the batch before it launches the sweeps (the rest is asynchronous on `stream` as ever)
 * and uses scratch of the engine: such calls on one engine go on one stream at a time.                        */
int valign_hip_set_ragged_batching(valign_hip_engine *e, int mode);

/* Score n pairs that are already in device memory: d_reads = n*read_length bytes and
 * d_refs = n*ref_length bytes (raw ASCII, pair-major, NUL padded), d_scores = n int16.
 * opt & 0xF: 0 Smith-Waterman, 1 Needleman-Wunsch variant; other values do nothing.
 * Asynchronous on `hip_stream` (a hipStream_t; NULL = the device's default stream).    */
int valign_hip_score_device(valign_hip_engine *e, int opt, long long n, const void *d_reads,
                            const void *d_refs, void *d_scores, void *hip_stream);

/* Align n device-resident pairs (linear gaps: the reference's model, Default-kernel tie-breaks;
 * affine scoring: the Gotoh extension, same tie-breaks where they apply): d_rows = n * 2 * (R+F) bytes,
 * per pair the read row then the ref row -- right-justified gapped strings in
 * [start, R+F-2], zeros before start, NUL at R+F-1 -- and d_idx = n * 4 int16
 * (readStart, readEnd, refStart, refEnd), i.e. the contents of the ABI's `Alignment`
 * (include/AlignmentKernel.h:12-18) flattened.  Tie-breaks follow the Default kernel.
 * Asynchronous on `hip_stream`; uses an internal pointer scratch (20.8 KB per pair at
 * 150x500, 25 MB at 10 kbp x 10 kbp; up to half the free HBM per launch -- at most 64 GiB for
 * reads of up to 2048 rows, 128 GiB for row strips --, larger batches run in chunks).                */
int valign_hip_align_device(valign_hip_engine *e, int opt, long long n, const void *d_reads,
                            const void *d_refs, void *d_rows, void *d_idx, void *hip_stream);

/* The plugin virtual without the C++ object: host pointers in, host scores out
 * (gather -> pinned staging -> H2D -> kernel -> D2H, chunked and overlapped).          */
int valign_hip_score_host(valign_hip_engine *e, int opt, int n, const char *const *reads,
                          const char *const *refs, short *scores, int threads);

/* compute_alignments for host pointers into caller-provided contiguous buffers: rows = n * 2 *
 * (read_length + ref_length) bytes (read row, then ref row, per pair; zeros before the start, NUL at
 * the end), idx = n * 4 shorts (readStart, readEnd, refStart, refEnd).  Same results as the plugin's
 * compute_alignments without its 2n operator new[] blocks -- for FFI callers (ctypes, cgo, JNI).      */
int valign_hip_align_host(valign_hip_engine *e, int opt, int n, const char *const *reads,
                          const char *const *refs, void *rows, short *idx, int threads);

/* Page-lock a host range and map it for the device (hipHostRegister behind a C symbol, so that an FFI caller needs no
 * HIP binding).  valign_hip_align_host into result buffers that lie inside a registered range -- or inside memory the
 * caller page-locked itself -- skips the library's pinned staging and its host-side copy: the device's copy engine
 * writes the caller's buffers directly (the flat layout IS the device layout).  Register once, reuse the buffers;
 * registering costs about a second per GB.  Unregister (with the pointer given to register) before freeing the memory.   */
int valign_hip_host_register(void *ptr, unsigned long long bytes);
int valign_hip_host_unregister(void *ptr);

/* Host-pointer score path (valign_hip_score_host / score_alignments): 1 (default) = the sequences cross PCIe as 4-bit
 * base classes, two per byte, and are expanded to one canonical byte per class in device memory -- the kernels only
 * ever look at the class of a base (DefaultKernel.h:43-60), so scores are identical; 0 = raw ASCII.  Plugin key:
 * host_packing.  Alignments always travel as the caller's bytes (they are copied into the result rows).                 */
int valign_hip_set_host_packing(valign_hip_engine *e, int mode);

/* score_alignments: 1 (default) = the recurrences run on packed half floats wherever every cell provably stays a small
 * integer (exact, bit-identical scores; v_pk_maximum3_f16 saves instructions), 0 = integer cells only.  Plugin key:
 * half_float_cells.                                                                                                      */
int valign_hip_set_half_float_cells(valign_hip_engine *e, int mode);

/* JSON description of what a call with this opt would launch (geometry, LDS, grid).   */
int valign_hip_describe(valign_hip_engine *e, int opt, long long n, char *buf, int cap);

const char *valign_hip_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* VALIGN_HIP_H */
