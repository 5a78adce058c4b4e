import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """Host harness + oracle are built on demand; the HIP plugin only if it is missing
    (the GPU box receives the prebuilt .so with the snapshot)."""
    from versalignlib_amd import build
    from oracle import cpu_ref
    build.build_host()
    cpu_ref.build()
    if not os.path.exists(build.HIP_PLUGIN):
        build.build_hip()
    yield


def ref_kernel(name):
    """Path of a reference kernel compiled by oracle/Makefile, or None when absent."""
    path = os.path.join(ROOT, "oracle", "_ref", "lib%sKernel.so" % name)
    return path if os.path.exists(path) else None


def band_constants():
    """(block_rows, col_align) of libHIPKernel.so's documented band, read from include/valign_hip.h."""
    import re
    text = open(os.path.join(ROOT, "include", "valign_hip.h")).read()
    return (int(re.search(r"#define\s+VALIGN_HIP_BAND_BLOCK_ROWS\s+(\d+)", text).group(1)),
            int(re.search(r"#define\s+VALIGN_HIP_BAND_COL_ALIGN\s+(\d+)", text).group(1)))
