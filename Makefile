# Convenience targets; the real build logic lives in versalignlib_amd/build.py (parallel hipcc) and
# oracle/Makefile (parity checker).
PY ?= python

all: build

build:
	$(PY) -c "import __graft_entry__ as g; g.build()"

test:
	$(PY) -m pytest tests -q -m "not gpu"

test-gpu:
	$(PY) -m pytest tests -q -m gpu

# CPU sanitizers: host_pipeline.h under TSan and ASan/UBSan, then the CPU suite against -fsanitize builds of
# libvalignhost.so, valign-bench and oracle/cpu_ref.c (tools/sanitize.sh)
sanitize:
	tools/sanitize.sh

smoke:
	$(PY) -c "import __graft_entry__ as g; g.smoke()"

bench:
	$(PY) bench.py

clean:
	rm -rf versalignlib_amd/build versalignlib_amd/lib oracle/libcpuref.so oracle/_ref

.PHONY: all build test test-gpu sanitize smoke bench clean
