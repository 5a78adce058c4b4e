"""In-tree builds of the two shared objects this package ships.

  lib/libHIPKernel.so .... the versalignLib plugin for MI355X (hipcc, gfx950 only)
  lib/libvalignhost.so ... the host side of the plugin protocol behind a flat C API (g++)

Both land under versalignlib_amd/lib/ (git-ignored, shipped to the GPU box by gpurun).
hipcc cross-compiles gfx950 without a GPU, so build_all() works on a CPU-only machine.
"""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "lib")
INCLUDE = os.path.join(ROOT, "include")

HIP_PLUGIN = os.path.join(LIB, "libHIPKernel.so")
HOST_LIB = os.path.join(LIB, "libvalignhost.so")
BENCH_CLI = os.path.join(LIB, "valign-bench")
# `make sanitize` (tools/sanitize.sh) builds the host-side pieces with -fsanitize=address,undefined into a directory
# of its own and points the test-suite at them; nothing is rebuilt from here then
SANITIZED_DIR = os.environ.get("VALIGN_SANITIZED_DIR") or None
if SANITIZED_DIR:
    HOST_LIB = os.path.join(SANITIZED_DIR, "libvalignhost.so")
    BENCH_CLI = os.path.join(SANITIZED_DIR, "valign-bench")

HIP_SOURCES = ["hip_plugin.hip", "engine_core.hip", "engine_score.hip", "engine_long.hip", "engine_align.hip"]
HIP_KERNEL_PART = "kernel_part.hip"          # compiled once per part (kernel_instances.hip.h), in parallel
HIP_KERNEL_PARTS = 6
# every header (a unit is rebuilt when its own source or any header changes; the kernel parts only look at KERNEL_PART_DEPS)
HIP_HEADERS = ["kernel_instances.hip.h", "dp_kernels.hip.h", "trace_kernels.hip.h", "long_kernels.hip.h", "strip_kernels.hip.h",
               "engine.hip.h", "host_runtime.hip.h", "host_pipeline.h", "band_kernels.hip.h", "pack_kernels.hip.h", "ragged_kernels.hip.h"]
OBJ = os.path.join(PKG, "build")             # intermediate objects (git-ignored)
HOST_SOURCES = ["valign_host.cpp"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _run(cmd):
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), proc.stdout))
    return proc.stdout


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP plugin cannot be built")


def build_host(force=False):
    if SANITIZED_DIR:
        if not os.path.exists(HOST_LIB):
            raise RuntimeError("VALIGN_SANITIZED_DIR is set but %s is missing: run tools/sanitize.sh" % HOST_LIB)
        return HOST_LIB
    os.makedirs(LIB, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in HOST_SOURCES]
    deps = srcs + [os.path.join(INCLUDE, h) for h in ("valign_host.h", "versalign_plugin_abi.h")]
    if force or _newer(HOST_LIB, deps):
        _run(["g++", "-std=c++14", "-O2", "-fPIC", "-shared", "-Wall", "-pthread",
              "-I" + INCLUDE] + srcs + ["-o", HOST_LIB, "-ldl"])
    cli_src = os.path.join(CSRC, "valign_bench.cpp")
    if force or _newer(BENCH_CLI, [cli_src, HOST_LIB]):
        _run(["g++", "-std=c++14", "-O2", "-Wall", "-I" + INCLUDE, cli_src, "-o", BENCH_CLI,
              "-L" + LIB, "-lvalignhost", "-Wl,-rpath,$ORIGIN", "-ldl", "-pthread"])
    return HOST_LIB


KERNEL_PART_DEPS = ["kernel_part.hip", "kernel_instances.hip.h", "dp_kernels.hip.h", "trace_kernels.hip.h"]


def build_hip(force=False, extra_flags=(), jobs=None):
    """hipcc -c per translation unit (the main one + HIP_KERNEL_PARTS kernel parts) in parallel, then
    one link.  A single hipcc run over all ~350 kernel instances takes minutes.  Units are rebuilt
    one by one: the kernel parts only include the per-geometry kernel headers (KERNEL_PART_DEPS), so
    a change to the engine or the plugin recompiles the main unit alone."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIB, exist_ok=True)
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(INCLUDE, h) for h in ("valign_hip.h", "versalign_plugin_abi.h")]
    all_deps = [os.path.join(CSRC, s) for s in HIP_HEADERS] + headers
    part_deps = [os.path.join(CSRC, s) for s in KERNEL_PART_DEPS]
    flags = list(extra_flags)
    stamp = os.path.join(OBJ, "flags.txt")             # objects built with other flags are stale
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(flags):
        force = True
    common = [hipcc_path(), "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-pthread", "-Wall", "-fvisibility=hidden", "-fvisibility-inlines-hidden",
              "--offload-compress",        # zstd-compressed code objects: 13.4 -> ~4 MB on disk, unpacked by the runtime at load
              "-Wno-unused-function", "-I" + INCLUDE, "-I" + CSRC] + flags
    units = [(os.path.join(CSRC, src), os.path.join(OBJ, os.path.splitext(src)[0] + ".o"), [], [os.path.join(CSRC, src)] + all_deps) for src in HIP_SOURCES]
    units += [(os.path.join(CSRC, HIP_KERNEL_PART), os.path.join(OBJ, "kernel_part%d.o" % i), ["-DVALIGN_PART=%d" % i], part_deps)
              for i in range(HIP_KERNEL_PARTS)]
    todo = [u for u in units if force or _newer(u[1], u[3])]
    if not todo and not _newer(HIP_PLUGIN, [u[1] for u in units]):
        return HIP_PLUGIN
    jobs = jobs or max(1, min(len(todo) or 1, os.cpu_count() or 1))
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        list(pool.map(lambda u: _run(common + u[2] + ["-c", u[0], "-o", u[1]]), todo))
    with open(stamp, "w") as f:
        f.write(" ".join(flags))
    _run([hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + [u[1] for u in units] + ["-o", HIP_PLUGIN])
    return HIP_PLUGIN


def build_all(force=False):
    return build_host(force), build_hip(force)


if __name__ == "__main__":
    import sys
    print("\n".join(build_all(force="--force" in sys.argv)))
