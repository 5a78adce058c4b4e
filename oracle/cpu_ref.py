"""ctypes view of oracle/libcpuref.so -- the CPU parity checker.

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never from versalignlib_amd (the product path).
The C source (oracle/cpu_ref.c) cites the reference lines it restates.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcpuref.so")
_SANITIZED = os.environ.get("VALIGN_SANITIZED_DIR") or None      # tools/sanitize.sh: the -fsanitize build of this file
if _SANITIZED:
    _LIB_PATH = os.path.join(_SANITIZED, "libcpuref.so")
REF_DIR = os.path.join(_HERE, "_ref")


class Scoring(ctypes.Structure):
    """Mirror of vref_scoring: linear model (reference) + affine extension."""

    _fields_ = [(k, ctypes.c_int32) for k in (
        "match", "mismatch", "gap_read", "gap_ref",
        "open_read", "ext_read", "open_ref", "ext_ref")]

    @classmethod
    def make(cls, match=2, mismatch=-1, gap_read=-3, gap_ref=-3,
             open_read=None, ext_read=None, open_ref=None, ext_ref=None):
        return cls(match, mismatch, gap_read, gap_ref,
                   gap_read if open_read is None else open_read,
                   gap_read if ext_read is None else ext_read,
                   gap_ref if open_ref is None else open_ref,
                   gap_ref if ext_ref is None else ext_ref)


def build(force=False):
    """Compile libcpuref.so (and oracle/_ref when the reference tree is present)."""
    if _SANITIZED:
        return
    stale = any(not os.path.exists(lib_path) or os.path.getmtime(lib_path) < os.path.getmtime(os.path.join(_HERE, src))
                for lib_path, src in ((_LIB_PATH, "cpu_ref.c"), (_SIMD_PATH, "cpu_simd.c")))
    if force or stale or not os.path.isdir(REF_DIR):
        subprocess.run(["make", "-C", _HERE, "--no-print-directory"], check=True,
                       stdout=subprocess.DEVNULL)


_SIMD_PATH = os.path.join(_SANITIZED or _HERE, "libcpusimd.so")
_simd = None


def simd_lib():
    """oracle/libcpusimd.so (cpu_simd.c): AVX2, 16 pairs per vector -- the SIMD CPU baseline of bench.py; None where
    the CPU has no AVX2."""
    global _simd
    if _simd is None:
        if not os.path.exists(_SIMD_PATH):
            build(force=True)
        L = ctypes.CDLL(_SIMD_PATH)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        L.vsimd_score.restype = ctypes.c_int
        L.vsimd_score.argtypes = [ctypes.c_int] * 5 + [u8p, u8p, ctypes.POINTER(Scoring), ctypes.POINTER(ctypes.c_int16), ctypes.c_int]
        L.vsimd_available.restype = ctypes.c_int
        _simd = L
    return _simd if _simd.vsimd_available() else None


def score_simd(opt, reads, refs, scoring=None, threads=1, affine=False):
    """The scores of score(...) from the AVX2 inter-sequence sweep (cpu_simd.c); raises where AVX2 is missing."""
    L = simd_lib()
    if L is None:
        raise RuntimeError("cpu_simd.c needs AVX2")
    reads, refs = _check(reads, refs)
    sc = scoring or Scoring.make()
    n, R = reads.shape
    F = refs.shape[1]
    out = np.zeros(n, dtype=np.int16)
    L.vsimd_score(opt, 1 if affine else 0, n, R, F, _u8(reads), _u8(refs), ctypes.byref(sc),
                  out.ctypes.data_as(ctypes.POINTER(ctypes.c_int16)), threads)
    return out


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        i16p = ctypes.POINTER(ctypes.c_int16)
        sp = ctypes.POINTER(Scoring)
        for name in ("vref_score", "vref_score_affine", "vref_score_wide", "vref_score_affine_wide"):
            fn = getattr(L, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_int] * 4 + [u8p, u8p, sp, i16p, ctypes.c_int]
        for name in ("vref_align", "vref_align_affine", "vref_align_sse", "vref_align_wide", "vref_align_affine_wide", "vref_align_sse_wide"):
            fn = getattr(L, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_int] * 4 + [u8p, u8p, sp, u8p, i16p, ctypes.c_int]
        for name in ("vref_score_banded_sw", "vref_score_banded_sw_affine"):
            fn = getattr(L, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_int] * 3 + [u8p, u8p, sp, ctypes.c_int, ctypes.c_int, ctypes.c_int, i16p, ctypes.c_int]
        L.vref_max_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _u8(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def _check(reads, refs):
    reads = np.ascontiguousarray(reads, dtype=np.uint8)
    refs = np.ascontiguousarray(refs, dtype=np.uint8)
    assert reads.ndim == 2 and refs.ndim == 2 and reads.shape[0] == refs.shape[0]
    return reads, refs


def score(opt, reads, refs, scoring=None, threads=1, affine=False, wide=False):
    """reads [n,R] uint8, refs [n,F] uint8 -> int16 [n] (full-width scores)."""
    reads, refs = _check(reads, refs)
    sc = scoring or Scoring.make()
    n, R = reads.shape
    F = refs.shape[1]
    out = np.zeros(n, dtype=np.int16)
    if affine:
        fn = lib().vref_score_affine_wide if wide else lib().vref_score_affine
    else:
        fn = lib().vref_score_wide if wide else lib().vref_score
    fn(opt, n, R, F, _u8(reads), _u8(refs), ctypes.byref(sc),
       out.ctypes.data_as(ctypes.POINTER(ctypes.c_int16)), threads)
    return out


def score_banded_sw(reads, refs, band_width, scoring=None, threads=1, block_rows=1, col_align=1, affine=False):
    """Banded Smith-Waterman scores (extension), band_width diagonals, 0 = every cell.  The default
    (block_rows = 1, col_align = 1) is the per-cell band |j - floor(i * F / R)| <= band_width / 2;
    libHIPKernel.so's documented band is block_rows = VALIGN_HIP_BAND_BLOCK_ROWS, col_align =
    VALIGN_HIP_BAND_COL_ALIGN (include/valign_hip.h), a superset of it."""
    reads, refs = _check(reads, refs)
    sc = scoring or Scoring.make()
    n, R = reads.shape
    F = refs.shape[1]
    out = np.zeros(n, dtype=np.int16)
    fn = lib().vref_score_banded_sw_affine if affine else lib().vref_score_banded_sw
    fn(n, R, F, _u8(reads), _u8(refs), ctypes.byref(sc), block_rows, col_align,
                               band_width // 2 if band_width > 0 else -1,
                               out.ctypes.data_as(ctypes.POINTER(ctypes.c_int16)), threads)
    return out


def align(opt, reads, refs, scoring=None, threads=1, affine=False, policy="default", wide=False):
    """-> rows uint8 [n,2,R+F] (zero before start, NUL at R+F-1), idx int16 [n,4].
    wide: int32 cells (every model and policy) -- identical wherever int16 does not overflow."""
    reads, refs = _check(reads, refs)
    sc = scoring or Scoring.make()
    n, R = reads.shape
    F = refs.shape[1]
    rows = np.zeros((n, 2, R + F), dtype=np.uint8)
    idx = np.zeros((n, 4), dtype=np.int16)
    L = lib()
    if affine:
        fn = L.vref_align_affine_wide if wide else L.vref_align_affine
    elif policy == "sse":
        fn = L.vref_align_sse_wide if wide else L.vref_align_sse
    else:
        fn = L.vref_align_wide if wide else L.vref_align
    fn(opt, n, R, F, _u8(reads), _u8(refs), ctypes.byref(sc), _u8(rows),
       idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int16)), threads)
    return rows, idx


def max_threads():
    return lib().vref_max_threads()
