"""versalignlib_amd -- an MI355X-native AlignmentKernel backend for versalignLib.

The product is lib/libHIPKernel.so: a versalignLib kernel plugin (four C symbols +
the AlignmentKernel virtuals, include/versalign_plugin_abi.h) whose Smith-Waterman /
Needleman-Wunsch DP runs as hand-written HIP on gfx950.  The Python modules here are
thin ctypes plumbing around it:

  build ....... in-tree hipcc / g++ builds
  host ........ the reference host protocol (dlopen any plugin by path)
  hipkernel ... the plugin's flat C API for device-resident batches
  synth ....... portable synthetic read/ref batches
  shard ....... contiguous pair-range sharding across ranks + score all-gather
"""
__version__ = "0.1.0"

# One HIP runtime per process: torch bundles its own libamdhip64 / libhsa-runtime64 with the
# same sonames as /opt/rocm's.  Whichever is loaded first serves everybody, and torch cannot
# initialise on top of the system copy, so when torch is installed let it load first.
try:  # pragma: no cover - depends on the environment
    import torch as _torch  # noqa: F401
except Exception:  # torch absent: the plugin and the host harness work without it
    _torch = None
