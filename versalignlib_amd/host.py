"""Python mirror of the versalignLib host protocol (ctypes over lib/libvalignhost.so).

`Plugin(path, **params)` does what the reference host does with a kernel library --
dlopen, set_parameters, set_logger, spawn_alignment_kernel (src/util/versalignUtil.cpp
:45-76, src/impl/main.cpp:227-238) -- and `score_alignments` / `compute_alignments`
call the two virtuals of include/AlignmentKernel.h:40-43.  Any versalignLib plugin works:
the reference's CPU kernels and this repo's libHIPKernel.so are swapped by path.
"""
import atexit
import ctypes
import os
import sys
import weakref

import numpy as np

from . import build as _build

SW = 0   # opt & 0xF == 0: Smith-Waterman
NW = 1   # opt & 0xF == 1: the reference's Needleman-Wunsch variant

_lib = None

# (as hipkernel.py: open plugins are closed by an atexit hook, not by finalizers at interpreter shutdown)
_live = weakref.WeakSet()


def _close_all():
    for obj in list(_live):
        try:
            obj.close()
        except Exception:
            pass


atexit.register(_close_all)


def lib():
    global _lib
    if _lib is None:
        path = _build.HOST_LIB
        if not os.path.exists(path):
            _build.build_host()
        L = ctypes.CDLL(path)
        vp = ctypes.c_void_p
        u8p = ctypes.POINTER(ctypes.c_uint8)
        i16p = ctypes.POINTER(ctypes.c_int16)
        L.vh_open.restype = vp
        L.vh_open.argtypes = [ctypes.c_char_p]
        L.vh_set_param.argtypes = [vp, ctypes.c_char_p, ctypes.c_int]
        L.vh_unset_param.argtypes = [vp, ctypes.c_char_p]
        L.vh_spawn.argtypes = [vp]
        L.vh_reapply_params.argtypes = [vp]
        L.vh_score.argtypes = [vp, ctypes.c_int, ctypes.c_int, u8p, u8p, i16p]
        L.vh_score_scattered.argtypes = [vp, ctypes.c_int, ctypes.c_int, u8p, u8p, i16p,
                                         ctypes.POINTER(ctypes.c_double)]
        L.vh_align.argtypes = [vp, ctypes.c_int, ctypes.c_int, u8p, u8p, u8p, i16p, ctypes.c_int]
        L.vh_time_calls.argtypes = [vp, ctypes.c_int, ctypes.c_int, u8p, u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.POINTER(ctypes.c_double)]
        L.vh_alloc_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        L.vh_last_call_seconds.restype = ctypes.c_double
        L.vh_last_call_seconds.argtypes = [vp]
        L.vh_close.restype = None
        L.vh_close.argtypes = [vp]
        L.vh_drain_log.argtypes = [vp, ctypes.c_char_p, ctypes.c_int]
        L.vh_log_to_stderr.restype = None
        L.vh_log_to_stderr.argtypes = [vp, ctypes.c_int]
        L.vh_last_error.restype = ctypes.c_char_p
        L.vh_parse_fasta.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p),
                                     ctypes.POINTER(ctypes.c_int)]
        L.vh_pad.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char,
                             ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int)]
        L.vh_cigar.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                               ctypes.c_char_p, ctypes.c_int]
        L.vh_free.restype = None
        L.vh_free.argtypes = [ctypes.c_void_p]
        _lib = L
    return _lib


class PluginError(RuntimeError):
    pass


def _err():
    return lib().vh_last_error().decode(errors="replace")


def _u8(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def _i16(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int16))


class Plugin:
    """One spawned AlignmentKernel of one plugin library.

    Parameter names are the reference's keys (src/impl/CustomParameters.h:9-24):
    score_match, score_mismatch, score_gap_read, score_gap_ref, read_length,
    ref_length, num_threads; further keys are passed through (libHIPKernel.so's
    optional extension keys).  A value of None removes a key, which is how the
    "lacking parameters" constructor error of the reference is exercised.
    """

    def __init__(self, so_path, read_length, ref_length, spawn=True, **params):
        self._h = lib().vh_open(os.fsencode(so_path))
        if not self._h:
            raise PluginError(_err())
        _live.add(self)
        self.read_length = int(read_length)
        self.ref_length = int(ref_length)
        self.set_params(read_length=read_length, ref_length=ref_length, **params)
        if spawn:
            self.spawn()

    def set_params(self, **params):
        for key, value in params.items():
            if value is None:
                lib().vh_unset_param(self._h, key.encode())
            else:
                lib().vh_set_param(self._h, key.encode(), int(value))
        if "read_length" in params and params["read_length"] is not None:
            self.read_length = int(params["read_length"])
        if "ref_length" in params and params["ref_length"] is not None:
            self.ref_length = int(params["ref_length"])

    def reapply_params(self):
        lib().vh_reapply_params(self._h)

    def spawn(self):
        if lib().vh_spawn(self._h) != 0:
            raise PluginError(_err())

    def _inputs(self, reads, refs):
        reads = np.ascontiguousarray(reads, dtype=np.uint8)
        refs = np.ascontiguousarray(refs, dtype=np.uint8)
        if reads.ndim != 2 or refs.ndim != 2 or reads.shape[0] != refs.shape[0]:
            raise ValueError("reads / refs must be [n, length] uint8 with equal n")
        if reads.shape[1] != self.read_length or refs.shape[1] != self.ref_length:
            raise ValueError("sequence lengths differ from read_length / ref_length")
        return reads, refs

    def score_alignments(self, opt, reads, refs, scattered=False):
        """-> int16 [n]; with scattered=True also the wall seconds of the virtual call."""
        reads, refs = self._inputs(reads, refs)
        n = reads.shape[0]
        scores = np.zeros(n, dtype=np.int16)
        if scattered:
            sec = ctypes.c_double(0.0)
            rc = lib().vh_score_scattered(self._h, opt, n, _u8(reads), _u8(refs), _i16(scores),
                                          ctypes.byref(sec))
            if rc != 0:
                raise PluginError(_err())
            return scores, sec.value
        if lib().vh_score(self._h, opt, n, _u8(reads), _u8(refs), _i16(scores)) != 0:
            raise PluginError(_err())
        return scores

    def compute_alignments(self, opt, reads, refs, normalise=True):
        """-> rows uint8 [n, 2, R+F] (read row, ref row), idx int16 [n, 4]."""
        reads, refs = self._inputs(reads, refs)
        n = reads.shape[0]
        rows = np.zeros((n, 2, self.read_length + self.ref_length), dtype=np.uint8)
        idx = np.zeros((n, 4), dtype=np.int16)
        rc = lib().vh_align(self._h, opt, n, _u8(reads), _u8(refs), _u8(rows), _i16(idx),
                            1 if normalise else 0)
        if rc != 0:
            raise PluginError(_err())
        return rows, idx

    def time_calls(self, opt, reads, refs, reps=100, align=True, free_between=False):
        """The reference host's timing loop (time_kernel, src/impl/main.cpp:268-292): `reps` back-to-back
        virtual calls on scattered heap blocks.  -> (seconds of the whole loop, [seconds per call])."""
        reads, refs = self._inputs(reads, refs)
        out = (ctypes.c_double * (reps + 1))()
        rc = lib().vh_time_calls(self._h, opt, reads.shape[0], _u8(reads), _u8(refs), int(reps), 1 if align else 0,
                                 1 if free_between else 0, out)
        if rc != 0:
            raise PluginError(_err())
        return out[0], list(out[1:])

    def last_call_seconds(self):
        """Wall seconds inside the plugin's last compute_alignments call."""
        return float(lib().vh_last_call_seconds(self._h))

    def drain_log(self):
        buf = ctypes.create_string_buffer(1 << 16)
        lib().vh_drain_log(self._h, buf, len(buf))
        return buf.value.decode(errors="replace")

    def log_to_stderr(self, on=True):
        lib().vh_log_to_stderr(self._h, 1 if on else 0)

    def close(self):
        if self._h:
            lib().vh_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        if sys is None or sys.is_finalizing():      # interpreter shutdown: the atexit hook has closed what was open
            return
        try:
            self.close()
        except Exception:
            pass


def alloc_probe(n, row_bytes, threads):
    """Seconds (allocate + fill, free) of 2 * n operator new[] rows of row_bytes on `threads` threads: what the
    ABI's result contract costs any backend on this host (include/AlignmentKernel.h:20-23)."""
    out = (ctypes.c_double * 2)()
    if lib().vh_alloc_probe(int(n), int(row_bytes), int(threads), out) != 0:
        raise PluginError(_err())
    return out[0], out[1]


def parse_fasta(path):
    """Sequences of a FASTA file as the reference host reads them -> list[bytes]."""
    blob = ctypes.c_void_p()
    count = ctypes.c_int(0)
    if lib().vh_parse_fasta(os.fsencode(path), ctypes.byref(blob), ctypes.byref(count)) != 0:
        raise PluginError(_err())
    out, off = [], 0
    base = blob.value
    for _ in range(count.value):
        s = ctypes.string_at(base + off)
        out.append(s)
        off += len(s) + 1
    lib().vh_free(blob)
    return out


def cigars(rows, idx, extended=False):
    """CIGAR strings of compute_alignments' output (rows uint8 [n, 2, R+F], idx int16 [n, 4]):
    M (or =/X), I = base only in the read, D = base only in the reference."""
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    n, AL = rows.shape[0], rows.shape[2]
    buf = ctypes.create_string_buffer(4 * AL + 16)
    out = []
    for i in range(n):
        base = rows.ctypes.data + i * 2 * AL
        got = lib().vh_cigar(base, base + AL, int(idx[i, 0]), int(idx[i, 1]), 1 if extended else 0,
                             buf, len(buf))
        if got < 0:
            raise PluginError(_err())
        out.append(buf.value.decode())
    return out


def pad(seqs, fill=b"\0"):
    """pad() of the reference host: right-pad to the longest -> uint8 [n, L]."""
    if not seqs:
        return np.zeros((0, 0), dtype=np.uint8)
    if any(b"\0" in bytes(s) for s in seqs):
        raise ValueError("sequences are C strings for pad(): embedded NUL bytes are not representable")
    blob = b"".join(bytes(s) + b"\0" for s in seqs)
    out = ctypes.c_void_p()
    length = ctypes.c_int(0)
    buf = ctypes.create_string_buffer(blob, len(blob))
    rc = lib().vh_pad(ctypes.cast(buf, ctypes.c_void_p), len(seqs), fill,
                      ctypes.byref(out), ctypes.byref(length))
    if rc != 0:
        raise PluginError(_err())
    n, L = len(seqs), length.value
    arr = np.ctypeslib.as_array(ctypes.cast(out, ctypes.POINTER(ctypes.c_uint8)),
                                shape=(n * L,)).copy().reshape(n, L) if n * L else \
        np.zeros((n, L), dtype=np.uint8)
    lib().vh_free(out)
    return arr
