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


def debug_switches(monkeypatch, **switches):
    """Set / clear entries of VALIGN_HIP_DEBUG -- the library's ONE environment variable, read when an engine is created
    ("name[=value],..."; tests and experiments only).  `name=None` removes an entry, `name=1` sets a flag."""
    current = {}
    for item in filter(None, os.environ.get("VALIGN_HIP_DEBUG", "").split(",")):
        name, _, value = item.partition("=")
        current[name] = value
    for name, value in switches.items():
        if value is None:
            current.pop(name, None)
        else:
            current[name] = str(int(value))
    if current:
        monkeypatch.setenv("VALIGN_HIP_DEBUG", ",".join("%s=%s" % kv for kv in sorted(current.items())))
    else:
        monkeypatch.delenv("VALIGN_HIP_DEBUG", raising=False)


def ref_kernel(name):
    """Path of a reference kernel compiled by oracle/Makefile, or None when absent."""
    path = os.path.join(ROOT, "oracle", "_ref", "lib%sKernel.so" % name)
    return path if os.path.exists(path) else None


def band_constants(R=None, F=None, band=0, scoring=None):
    """(block_rows, col_align) of the band libHIPKernel.so computes for this shape / band / scoring, as the library
    itself reports them (valign_hip_describe: "band_block_rows", "band_col_align"; include/valign_hip.h documents
    which block shape applies where).  Needs the GPU library; without a shape: the header's strip constants."""
    import re
    if R is None:
        text = open(os.path.join(ROOT, "include", "valign_hip.h")).read()
        return (int(re.search(r"#define\s+VALIGN_HIP_BAND_BLOCK_ROWS\s+(\d+)", text).group(1)),
                int(re.search(r"#define\s+VALIGN_HIP_BAND_COL_ALIGN\s+(\d+)", text).group(1)))
    from versalignlib_amd import hipkernel
    eng = hipkernel.Engine(R, F, scoring)
    eng.set_band_width(band)
    d = eng.describe(0, 1)
    eng.close()
    return d["band_block_rows"], d["band_col_align"]
