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

smoke:
	$(PY) -c "import __graft_entry__ as g; g.smoke()"

bench:
	$(PY) bench.py

clean:
	rm -rf versalignlib_amd/build versalignlib_amd/lib oracle/libcpuref.so oracle/_ref

.PHONY: all build test test-gpu smoke bench clean
