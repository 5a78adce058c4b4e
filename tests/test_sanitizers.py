"""The host-only half of the plugin's host-pointer path (versalignlib_amd/csrc/host_pipeline.h: the worker pool and
the gather / scatter that up to 64 host threads run into caller-owned arrays) under ThreadSanitizer and under
AddressSanitizer + UBSan, on the CPU (the reference shipped a data race in exactly this place,
src/Kernels/AVX-SSE/SSEKernel.cpp:77-82, and the GPU pool offers no sanitizers).  tests/host_pipeline_check.cpp is
the driver; `tools/sanitize.sh` additionally runs this whole CPU suite against sanitized builds of libvalignhost.so,
valign-bench and the oracle."""
import os
import subprocess

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "host_pipeline_check.cpp")
CSRC = os.path.join(ROOT, "versalignlib_amd", "csrc")


@pytest.mark.parametrize("name,flags,env", [
    ("tsan", ["-fsanitize=thread"], {"TSAN_OPTIONS": "halt_on_error=1"}),
    ("asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], {"ASAN_OPTIONS": "detect_leaks=1"}),
])
def test_host_pipeline_under_sanitizers(tmp_path, name, flags, env):
    exe = str(tmp_path / ("host_pipeline_" + name))
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-pthread", "-I" + CSRC] + flags +
                           [SRC, "-o", exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert build.returncode == 0, build.stdout[-3000:]
    run_env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}      # (the program carries its own runtime)
    run_env.update(env)
    res = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=run_env)
    assert res.returncode == 0 and "host pipeline ok" in res.stdout, res.stdout[-3000:]
    assert "WARNING: ThreadSanitizer" not in res.stdout and "ERROR: AddressSanitizer" not in res.stdout
    assert "runtime error" not in res.stdout


def test_engine_uses_the_tested_header():
    """The engine must run THIS code (not a private copy of it) on its host threads."""
    text = open(os.path.join(CSRC, "engine.hip.h")).read()
    assert '#include "host_pipeline.h"' in text
    assert "class WorkerPool" not in text and "packer_.gather(" in text and "packer_.scatter(" in text
    for unit in ("engine_core.hip", "engine_score.hip", "engine_long.hip", "engine_align.hip"):
        assert "class WorkerPool" not in open(os.path.join(CSRC, unit)).read()
