/*
 * valign_host.h -- flat C view of the versalignLib HOST side (libvalignhost.so).
 *
 * The reference host (src/impl/main.cpp, src/util/versalignUtil.cpp) is C++ that
 * dlopen()s a kernel plugin, injects a parameter object and a logger, spawns the
 * kernel and calls its two virtuals.  This library speaks exactly that protocol and
 * exposes it through plain C entry points (pointers and sizes only), so that tests,
 * bench.py and any FFI can drive ANY versalignLib plugin by path -- the reference's
 * libDefaultKernel.so / libSSEKernel.so / libAVXKernel.so and this repo's
 * libHIPKernel.so are interchangeable behind it.
 *
 * Reference counterparts:
 *   vh_open / vh_spawn ....... DLL_init + get_kernel   (src/util/versalignUtil.cpp:45-76,
 *                                                       src/impl/main.cpp:227-238)
 *   vh_set_param ............. CustomParameters fields (src/impl/CustomParameters.h:9-58)
 *   vh_score / vh_align ...... kernel->score_alignments / compute_alignments
 *                                                      (src/impl/main.cpp:131,143)
 *   vh_close ................. clear_kernel            (src/impl/main.cpp:217-225)
 *   vh_pad ................... pad()                   (src/util/versalignUtil.cpp:17-33)
 *   vh_parse_fasta ........... FastaProvider           (src/util/versalignUtil.h:52-92)
 *   vh_cigar ................. (none; the rows main.cpp:147-189 prints, run-length encoded)
 *
 * All functions return 0 on success and a negative value on failure unless stated
 * otherwise; vh_last_error() gives the message of the most recent failure on the
 * calling thread.
 */
#ifndef VALIGN_HOST_H
#define VALIGN_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vh_plugin vh_plugin;

/* dlopen(path, RTLD_LAZY) and resolve the four plugin symbols.  Nothing is spawned
 * yet.  Parameter defaults are the reference's: score_match 2, score_mismatch -1,
 * score_gap_read -3, score_gap_ref -3, num_threads 1 (CustomParameters.h:49-58).   */
vh_plugin *vh_open(const char *so_path);

/* Set / override one integer key of the parameter object handed to the plugin.
 * Unknown keys become known (has_key -> true), which is how the optional extension
 * keys of libHIPKernel.so are supplied.  vh_unset_param removes a key again.       */
int vh_set_param(vh_plugin *p, const char *key, int value);
int vh_unset_param(vh_plugin *p, const char *key);

/* set_parameters + set_logger + spawn_alignment_kernel.  A plugin constructor that
 * throws (the reference throws `const char *` when a required key is missing,
 * DefaultKernel.h:79-81) is reported as an error, not propagated.                   */
int vh_spawn(vh_plugin *p);

/* Re-inject the parameter object (reference host does this before every timing
 * trial, main.cpp:263).                                                              */
int vh_reapply_params(vh_plugin *p);

/* reads = n * read_length bytes, refs = n * ref_length bytes, pair-major.  The
 * harness builds the char** arrays the ABI wants (pointing into these buffers) and
 * calls score_alignments.  scores must hold n shorts.                               */
int vh_score(vh_plugin *p, int opt, int n, const uint8_t *reads, const uint8_t *refs,
             int16_t *scores);

/* compute_alignments.  rows = n * 2 * (R+F) bytes (read row, then ref row);
 * idx = n * 4 shorts (readStart, readEnd, refStart, refEnd).  With normalise != 0
 * bytes outside [readStart, R+F-2] are zeroed in the copy (the Default kernel
 * leaves them uninitialised for Smith-Waterman).  The Alignment objects are
 * destroyed (rows delete[]d) before returning.                                       */
int vh_align(vh_plugin *p, int opt, int n, const uint8_t *reads, const uint8_t *refs,
             uint8_t *rows, int16_t *idx, int normalise);

/* Wall seconds the plugin spent inside the last compute_alignments virtual call of vh_align
 * (without the harness's own copy-out and the Alignment destructors).                         */
double vh_last_call_seconds(vh_plugin *p);

/* As vh_score / vh_align but every sequence is first copied into its own heap block
 * (what the reference host's pad() produces), so gather costs are realistic; used
 * for end-to-end timing.  seconds_out receives the wall time of the virtual call.   */
int vh_score_scattered(vh_plugin *p, int opt, int n, const uint8_t *reads,
                       const uint8_t *refs, int16_t *scores, double *seconds_out);

/* The reference host's timing protocol (time_kernel, src/impl/main.cpp:268-292): `reps` back-to-back
 * calls of compute_alignments (align != 0) or score_alignments on the same n pairs, every sequence in
 * its own heap block (pad()), one timer around the whole loop.  The reference leaks the Alignment rows
 * of every repetition (main.cpp:280-285); here every repetition gets its own value-initialised
 * Alignment array and all of them are destroyed after the timer has stopped, so the timed region does
 * the same work.  seconds_out[0] = the whole loop, seconds_out[1 + r] = repetition r (reps + 1 doubles).
 * free_between != 0: the rows of one repetition are freed before the next starts, as a host that
 * consumes its results would (the timed regions still exclude the frees).                             */
int vh_time_calls(vh_plugin *p, int opt, int n, const uint8_t *reads, const uint8_t *refs, int reps,
                  int align, int free_between, double *seconds_out);

/* What the ABI's result contract costs ANY backend on this host: `threads` threads allocate 2 * n rows of
 * row_bytes with operator new[] (the caller's ~Alignment delete[]s them, include/AlignmentKernel.h:20-23) and
 * write every byte once -- no plugin involved.  seconds_out[0] = the allocation + fill, seconds_out[1] = the
 * delete[]s by the calling thread.  A measuring aid for bench.py's `abi` object.                            */
int vh_alloc_probe(int n, int row_bytes, int threads, double *seconds_out);

/* delete_alignment_kernel + dlclose + free.                                          */
void vh_close(vh_plugin *p);

/* Lines the plugin sent to the injected logger since the last call (newline
 * separated, truncated to cap-1 bytes).  Returns the number of bytes written.       */
int vh_drain_log(vh_plugin *p, char *buf, int cap);
/* 0 = keep log lines in memory only (default), 1 = also print them to stderr in
 * the reference format "SEVERITY\t[module]\tmessage".                                */
void vh_log_to_stderr(vh_plugin *p, int on);

const char *vh_last_error(void);

/* ---- host data formats (callers either side of the hot path) ---- */

/* Parse a FASTA file the way the reference does (records whose sequence lines
 * contain a blank are dropped).  On success *count sequences are returned in one
 * malloc'ed blob of NUL-terminated strings laid end to end; free with vh_free.      */
int vh_parse_fasta(const char *path, char **blob, int *count);

/* Right-pad `count` NUL-terminated strings (laid end to end in blob) with `fill` to
 * the longest length; writes count * (*length) bytes into a malloc'ed *out.         */
int vh_pad(const char *blob, int count, char fill, uint8_t **out, int *length);

/* CIGAR of one alignment as compute_alignments returns it (two gapped rows of read_length +
 * ref_length bytes, Alignment::readStart .. readEnd-1 are the used columns, the last one the NUL
 * the kernels write): a column with a base in both rows is M (extended != 0: '=' where the two
 * bytes are equal ignoring case, else 'X'), a '-' in the ref row is I (base only in the read), a '-'
 * in the read row is D.  The reference has no CIGAR writer -- its host prints the two rows
 * (src/impl/main.cpp:147-189); this is the same information run-length encoded.  Writes a
 * NUL-terminated string into buf; returns its length, or -1 when cap is too small / bad input.
 * An empty alignment gives "".                                                                   */
int vh_cigar(const uint8_t *read_row, const uint8_t *ref_row, int start, int end, int extended,
             char *buf, int cap);

void vh_free(void *ptr);

#ifdef __cplusplus
}
#endif
#endif /* VALIGN_HOST_H */
