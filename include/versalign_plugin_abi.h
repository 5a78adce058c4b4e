/*
 * versalign_plugin_abi.h -- the versalignLib kernel-plugin boundary, restated.
 *
 * A versalignLib kernel backend is a shared object that a host dlopen()s and talks
 * to through (a) four unmangled C symbols and (b) three tiny C++ interface types
 * whose objects cross the boundary by pointer (Itanium C++ ABI: what matters is the
 * data layout of `Alignment` and the ORDER of the virtual functions, which fixes the
 * vtable slots).  This header restates those three types so that libHIPKernel.so and
 * this repo's host harness compile without the reference tree; a host built against
 * the reference's own headers and a plugin built against this file interoperate.
 *
 * Reference interfaces restated here (paths relative to the reference checkout):
 *   struct Alignment ............ include/AlignmentKernel.h:12-24
 *   class AlignmentKernel ....... include/AlignmentKernel.h:34-44
 *   factory typedefs ............ include/AlignmentKernel.h:46-47
 *   class AlignmentParameters ... include/AlignmentParameters.h:11-17, typedef :19
 *   class AlignmentLogger ....... include/AlignmentLogger.h:13-17,   typedef :19
 *   plugin-global pointers ...... include/AlignmentParameters.h:21-22,
 *                                 include/AlignmentLogger.h:21-22
 *
 * Each section is wrapped in the include guard the reference uses for the matching
 * header, so a translation unit may include either set (or both, in any order)
 * without redefinitions.
 */
#ifndef VERSALIGN_PLUGIN_ABI_H
#define VERSALIGN_PLUGIN_ABI_H

#include <stddef.h>
#include <stdio.h>

/* ------------------------------------------------------------------ results */
#ifndef ALIGNMENTKERNEL_H
#define ALIGNMENTKERNEL_H

/* One alignment result, 24 bytes: two heap rows of read_length + ref_length chars,
 * right-justified (the gapped strings occupy [readStart, readEnd - 1], byte readEnd
 * is '\0'), plus four 16-bit coordinates.  The HOST's destructor delete[]s both rows,
 * so a plugin must allocate them with operator new[] from the shared C++ runtime.   */
struct Alignment {
    char *read = 0;
    char *ref = 0;
    short readStart;
    short readEnd;
    short refStart;
    short refEnd;

    ~Alignment() {
        if (read != 0) delete[] read;
        if (ref != 0) delete[] ref;
    }
};

/* The kernel object a plugin hands out.  `opt & 0xF` selects the algorithm:
 * 0 = Smith-Waterman, 1 = the reference's Needleman-Wunsch variant; any other value
 * is a silent no-op (src/Kernels/default/DefaultKernel.cpp:24-40,55-69).
 * reads[i] / refs[i] point to exactly read_length / ref_length bytes (no terminator).
 * Vtable slots, in order: destructor (2), score_alignments, compute_alignments.      */
class AlignmentKernel {
public:
    virtual ~AlignmentKernel() {}

    virtual void score_alignments(int const &opt, int const &aln_number,
                                  char const *const *const reads,
                                  char const *const *const refs,
                                  short *const scores) = 0;

    virtual void compute_alignments(int const &opt, int const &aln_number,
                                    char const *const *const reads,
                                    char const *const *const refs,
                                    Alignment *const alignments) = 0;
};

typedef AlignmentKernel *(*fp_load_alignment_kernel)();
typedef void (*fp_delete_alignment_kernel)(AlignmentKernel *);

#endif /* ALIGNMENTKERNEL_H */

/* --------------------------------------------------------------- parameters */
#ifndef INCLUDE_ALIGNMENTPARAMETERS_H
#define INCLUDE_ALIGNMENTPARAMETERS_H

/* Key -> int configuration object owned by the host.  Vtable slots, in order:
 * param_int, has_key, destructor (2).  Keys every reference backend requires at
 * construction: score_match, score_mismatch, score_gap_read, score_gap_ref,
 * read_length, ref_length; num_threads is read per call
 * (src/Kernels/default/DefaultKernel.h:70-81, DefaultKernel.cpp:45).                 */
class AlignmentParameters {
public:
    virtual int param_int(char const *const key) = 0;
    virtual bool has_key(char const *const key) = 0;

    virtual ~AlignmentParameters() {}
};

typedef void (*fp_set_parameters)(AlignmentParameters const *);

/* Defined once inside each plugin; private to it (hosts dlopen without RTLD_GLOBAL). */
extern AlignmentParameters *_parameters;
#define Parameters (*_parameters)

#endif /* INCLUDE_ALIGNMENTPARAMETERS_H */

/* ------------------------------------------------------------------- logger */
#ifndef ALIGNMENTLOGGER_H
#define ALIGNMENTLOGGER_H

/* Host-owned log sink.  level: 0 INFO, 1 WARNING, 3 DRASTIC, anything else ERROR
 * (src/impl/CustomLogger.h:22-35).  `arg_num` extra `char const *` lines may follow.
 * Host implementations are not thread-safe: call from the thread that entered the
 * plugin only.  Vtable slots, in order: log, destructor (2).                         */
class AlignmentLogger {
public:
    virtual void log(int const level, char const *const main, char const *const msg,
                     size_t const &arg_num = 0, ...) = 0;
    virtual ~AlignmentLogger() {}
};

typedef void (*fp_set_logger)(AlignmentLogger const *);

extern AlignmentLogger *_logger;
#define Logger (*_logger)

#endif /* ALIGNMENTLOGGER_H */

/* ---------------------------------------------- the four exported C symbols */
/* Every plugin exports exactly these (src/Kernels/default/DefaultKernel_dllexport.cpp
 * :18-42, identical in the SSE/AVX/OpenCL backends).  Call order used by the
 * reference host: dlopen -> set_parameters -> set_logger
 * (src/util/versalignUtil.cpp:45-76) -> spawn_alignment_kernel (src/impl/main.cpp
 * :227-238) -> score_alignments / compute_alignments ... -> delete_alignment_kernel.
 * set_parameters may be called again at any time (src/impl/main.cpp:259-263).        */
#define VERSALIGN_SYM_SPAWN      "spawn_alignment_kernel"
#define VERSALIGN_SYM_DELETE     "delete_alignment_kernel"
#define VERSALIGN_SYM_SET_PARAMS "set_parameters"
#define VERSALIGN_SYM_SET_LOGGER "set_logger"

#endif /* VERSALIGN_PLUGIN_ABI_H */
