"""ctypes plumbing over lib/libHIPKernel.so's flat C API (include/valign_hip.h).

`Engine` is the device-resident twin of a spawned kernel object: fixed
(read_length, ref_length, scoring), batches passed as device pointers.  torch is used
only to own device memory and streams.  There is no CPU fallback: constructing an
Engine without a gfx950 device raises.
"""
import atexit
import ctypes
import json
import os
import sys
import weakref

from . import build as _build

SW = 0
NW = 1

_lib = None

# Objects that own a handle of the native library.  They are closed by an atexit hook -- while the interpreter, ctypes and
# the HIP runtime are all still whole -- instead of by finalizers that the interpreter's shutdown runs in no particular
# order (a finalizer that frees device memory after the runtime's own teardown ends the process with an abort).
_live = weakref.WeakSet()


def _track(obj):
    _live.add(obj)


def _close_all():
    for obj in list(_live):
        try:
            obj.close()
        except Exception:
            pass


atexit.register(_close_all)


class HipKernelError(RuntimeError):
    pass


class Scoring(ctypes.Structure):
    """Mirror of valign_hip_scoring."""

    _fields_ = [(k, ctypes.c_int32) for k in (
        "match", "mismatch", "gap_read", "gap_ref", "affine",
        "open_read", "ext_read", "open_ref", "ext_ref")]

    @classmethod
    def make(cls, match=2, mismatch=-1, gap_read=-3, gap_ref=-3,
             open_read=None, ext_read=None, open_ref=None, ext_ref=None):
        affine = any(v is not None for v in (open_read, ext_read, open_ref, ext_ref))
        return cls(match, mismatch, gap_read, gap_ref, 1 if affine else 0,
                   gap_read if open_read is None else open_read,
                   gap_read if ext_read is None else ext_read,
                   gap_ref if open_ref is None else open_ref,
                   gap_ref if ext_ref is None else ext_ref)


def plugin_path():
    return _build.HIP_PLUGIN


def lib():
    """Load libHIPKernel.so; raises if it has not been built (never falls back)."""
    global _lib
    if _lib is None:
        path = plugin_path()
        if not os.path.exists(path):
            raise HipKernelError(
                "libHIPKernel.so is missing (%s): run `python -m versalignlib_amd.build`" % path)
        L = ctypes.CDLL(path)
        vp = ctypes.c_void_p
        L.valign_hip_device_count.restype = ctypes.c_int
        L.valign_hip_engine_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.POINTER(Scoring), ctypes.c_int, ctypes.c_int,
                                               ctypes.POINTER(vp)]
        L.valign_hip_engine_destroy.restype = None
        L.valign_hip_engine_destroy.argtypes = [vp]
        L.valign_hip_set_traceback_policy.argtypes = [vp, ctypes.c_int]
        L.valign_hip_set_band_width.argtypes = [vp, ctypes.c_int]
        L.valign_hip_set_pointer_scratch_cap_mb.argtypes = [vp, ctypes.c_longlong]
        L.valign_hip_set_host_packing.argtypes = [vp, ctypes.c_int]
        L.valign_hip_set_half_float_cells.argtypes = [vp, ctypes.c_int]
        L.valign_hip_host_register.argtypes = [vp, ctypes.c_ulonglong]
        L.valign_hip_host_unregister.argtypes = [vp]
        L.valign_hip_set_score_width.argtypes = [vp, ctypes.c_int]
        L.valign_hip_set_ragged_batching.argtypes = [vp, ctypes.c_int]
        L.valign_hip_score_device.argtypes = [vp, ctypes.c_int, ctypes.c_longlong, vp, vp, vp, vp]
        L.valign_hip_align_device.argtypes = [vp, ctypes.c_int, ctypes.c_longlong, vp, vp, vp, vp, vp]
        L.valign_hip_score_host.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_int]
        L.valign_hip_align_host.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, ctypes.c_int]
        L.valign_hip_describe.argtypes = [vp, ctypes.c_int, ctypes.c_longlong, ctypes.c_char_p,
                                          ctypes.c_int]
        L.valign_hip_last_error.restype = ctypes.c_char_p
        _lib = L
    return _lib


EXPORTED_SYMBOLS = (
    "spawn_alignment_kernel", "set_parameters", "set_logger", "delete_alignment_kernel",
    "valign_hip_device_count", "valign_hip_shard_range", "valign_hip_engine_create", "valign_hip_engine_destroy",
    "valign_hip_set_traceback_policy", "valign_hip_set_pointer_scratch_cap_mb", "valign_hip_set_host_packing", "valign_hip_set_half_float_cells", "valign_hip_host_register", "valign_hip_host_unregister", "valign_hip_set_band_width", "valign_hip_set_score_width", "valign_hip_set_ragged_batching", "valign_hip_score_device", "valign_hip_align_device", "valign_hip_score_host", "valign_hip_align_host", "valign_hip_describe",
    "valign_hip_last_error",
)


def _err():
    return lib().valign_hip_last_error().decode(errors="replace")


def host_register(array):
    """Page-lock a numpy array for the device (valign_hip_host_register): result buffers registered once take the
    device's copies directly in Engine.align_host(out=...).  Unregister before the array is freed."""
    if lib().valign_hip_host_register(array.ctypes.data, array.nbytes) != 0:
        raise HipKernelError(_err())


def host_unregister(array):
    if lib().valign_hip_host_unregister(array.ctypes.data) != 0:
        raise HipKernelError(_err())


class Engine:
    def __init__(self, read_length, ref_length, scoring=None, device=0, group_lanes=0,
                 rows_per_lane=0):
        self.read_length = int(read_length)
        self.ref_length = int(ref_length)
        self.scoring = scoring or Scoring.make()
        self.device = int(device)
        self._h = ctypes.c_void_p()
        rc = lib().valign_hip_engine_create(self.device, self.read_length, self.ref_length,
                                            ctypes.byref(self.scoring), int(group_lanes),
                                            int(rows_per_lane), ctypes.byref(self._h))
        if rc != 0:
            self._h = None
            raise HipKernelError(_err())
        _track(self)

    def set_traceback_policy(self, policy):
        """0: Default-kernel tie-breaks (default); 1: SSE/AVX-kernel tie-breaks."""
        if lib().valign_hip_set_traceback_policy(self._h, int(policy)) != 0:
            raise HipKernelError(_err())

    def set_pointer_scratch_cap_mb(self, mb):
        """Cap of compute_alignments' device-side pointer scratch in MiB (0: 64 GiB / half the free HBM)."""
        if lib().valign_hip_set_pointer_scratch_cap_mb(self._h, int(mb)) != 0:
            raise HipKernelError(_err())

    def set_host_packing(self, mode):
        """Host-pointer score path: 1 = 4-bit base classes across PCIe (default), 0 = raw ASCII."""
        if lib().valign_hip_set_host_packing(self._h, int(mode)) != 0:
            raise HipKernelError(_err())

    def set_half_float_cells(self, mode):
        """score path: 1 = half-float cells where exact (default), 0 = integer cells only (identical scores)."""
        if lib().valign_hip_set_half_float_cells(self._h, int(mode)) != 0:
            raise HipKernelError(_err())

    def set_score_width(self, bits):
        """0 auto (int16, int32 where needed), 16, or 32."""
        if lib().valign_hip_set_score_width(self._h, int(bits)) != 0:
            raise HipKernelError(_err())

    def set_ragged_batching(self, mode):
        """Length-sorted score batches (classified, packed and swept by length class on the device; host-pointer calls
        and score_device): 0 never (default), 1 when the call is ragged enough, 2 always."""
        if lib().valign_hip_set_ragged_batching(self._h, int(mode)) != 0:
            raise HipKernelError(_err())

    def set_band_width(self, diagonals):
        """> 0: banded Smith-Waterman scores (strip band); 0: every cell."""
        if lib().valign_hip_set_band_width(self._h, int(diagonals)) != 0:
            raise HipKernelError(_err())

    def score_device(self, opt, reads, refs, scores=None, stream=None):
        """reads/refs: torch uint8 CUDA tensors [n, R] / [n, F]; -> int16 CUDA tensor [n].

        Asynchronous on torch's current stream (or `stream`)."""
        import torch
        n = reads.shape[0]
        assert reads.is_cuda and refs.is_cuda and reads.dtype == torch.uint8 and refs.dtype == torch.uint8
        assert reads.is_contiguous() and refs.is_contiguous()
        assert tuple(reads.shape) == (n, self.read_length) and tuple(refs.shape) == (n, self.ref_length)
        if scores is None:
            scores = torch.empty(n, dtype=torch.int16, device=reads.device)
        st = stream if stream is not None else torch.cuda.current_stream(reads.device)
        rc = lib().valign_hip_score_device(self._h, int(opt), n, reads.data_ptr(), refs.data_ptr(),
                                           scores.data_ptr(), st.cuda_stream)
        if rc != 0:
            raise HipKernelError(_err())
        return scores

    def align_device(self, opt, reads, refs, rows=None, idx=None, stream=None):
        """-> rows uint8 CUDA [n, 2, R+F], idx int16 CUDA [n, 4] (async on the current stream)."""
        import torch
        n = reads.shape[0]
        assert reads.is_cuda and refs.is_cuda and reads.is_contiguous() and refs.is_contiguous()
        assert tuple(reads.shape) == (n, self.read_length) and tuple(refs.shape) == (n, self.ref_length)
        AL = self.read_length + self.ref_length
        if rows is None:
            rows = torch.empty((n, 2, AL), dtype=torch.uint8, device=reads.device)
        if idx is None:
            idx = torch.empty((n, 4), dtype=torch.int16, device=reads.device)
        st = stream if stream is not None else torch.cuda.current_stream(reads.device)
        rc = lib().valign_hip_align_device(self._h, int(opt), n, reads.data_ptr(), refs.data_ptr(),
                                           rows.data_ptr(), idx.data_ptr(), st.cuda_stream)
        if rc != 0:
            raise HipKernelError(_err())
        return rows, idx

    def score_host(self, opt, reads, refs, threads=1):
        """Host-pointer path of score_alignments without the plugin object: reads/refs are C-contiguous
        numpy uint8 arrays [n, R] / [n, F]; the library gets one pointer per sequence."""
        import numpy as np
        n = reads.shape[0]
        assert reads.dtype == np.uint8 and refs.dtype == np.uint8
        assert reads.flags.c_contiguous and refs.flags.c_contiguous
        assert tuple(reads.shape) == (n, self.read_length) and tuple(refs.shape) == (n, self.ref_length)
        rp = (reads.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(self.read_length)).astype(np.uint64)
        fp = (refs.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(self.ref_length)).astype(np.uint64)
        scores = np.zeros(n, dtype=np.int16)
        rc = lib().valign_hip_score_host(self._h, int(opt), n, rp.ctypes.data, fp.ctypes.data,
                                         scores.ctypes.data, int(threads))
        if rc != 0:
            raise HipKernelError(_err())
        return scores

    def align_host(self, opt, reads, refs, threads=1, out=None):
        """Host-pointer path of compute_alignments into contiguous numpy buffers (no operator new[] blocks):
        -> rows uint8 [n, 2, R+F], idx int16 [n, 4].  `out` = (rows, idx) of an earlier call is written in place
        (a caller that loops keeps its result buffers, like any FFI binding would)."""
        import numpy as np
        n = reads.shape[0]
        assert reads.dtype == np.uint8 and refs.dtype == np.uint8 and reads.flags.c_contiguous and refs.flags.c_contiguous
        assert tuple(reads.shape) == (n, self.read_length) and tuple(refs.shape) == (n, self.ref_length)
        rp = (reads.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(self.read_length)).astype(np.uint64)
        fp = (refs.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(self.ref_length)).astype(np.uint64)
        AL = self.read_length + self.ref_length
        if out is not None:
            rows, idx = out
            assert rows.shape == (n, 2, AL) and rows.dtype == np.uint8 and rows.flags.c_contiguous
            assert idx.shape == (n, 4) and idx.dtype == np.int16 and idx.flags.c_contiguous
        else:
            rows = np.zeros((n, 2, AL), dtype=np.uint8)
            idx = np.zeros((n, 4), dtype=np.int16)
        rc = lib().valign_hip_align_host(self._h, int(opt), n, rp.ctypes.data, fp.ctypes.data, rows.ctypes.data,
                                         idx.ctypes.data, int(threads))
        if rc != 0:
            raise HipKernelError(_err())
        return rows, idx

    def describe(self, opt=0, n=0):
        buf = ctypes.create_string_buffer(2048)
        lib().valign_hip_describe(self._h, int(opt), int(n), buf, len(buf))
        return json.loads(buf.value.decode())

    def close(self):
        if getattr(self, "_h", None):
            lib().valign_hip_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        if sys is None or sys.is_finalizing():      # interpreter shutdown: the atexit hook has closed what was open
            return
        try:
            self.close()
        except Exception:
            pass
