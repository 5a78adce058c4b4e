#!/bin/bash
# CPU-side sanitizer run ("race detection / sanitizers"; the GPU pool offers none):
#   1. the host-only half of the plugin's host-pointer path (versalignlib_amd/csrc/host_pipeline.h: worker pool,
#      gather, scatter) as a stand-alone program under -fsanitize=thread and under -fsanitize=address,undefined;
#   2. libvalignhost.so, valign-bench and the oracle (oracle/cpu_ref.c) built with -fsanitize=address,undefined
#      into build/sanitize/, and the whole CPU test-suite run against THOSE (python gets the runtimes preloaded).
# Usage: tools/sanitize.sh [pytest args...]      (also: make sanitize)
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/build/sanitize
mkdir -p "$OUT"
CS=$R/versalignlib_amd/csrc
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1"

echo "== host_pipeline.h under ThreadSanitizer"
g++ -std=c++17 -O1 -g -fsanitize=thread -fno-omit-frame-pointer -pthread -I"$CS" "$R/tests/host_pipeline_check.cpp" -o "$OUT/host_pipeline_tsan"
TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1" "$OUT/host_pipeline_tsan"
echo "== host_pipeline.h under AddressSanitizer + UBSan"
g++ -std=c++17 $SAN -pthread -I"$CS" "$R/tests/host_pipeline_check.cpp" -o "$OUT/host_pipeline_asan"
ASAN_OPTIONS="detect_leaks=1" "$OUT/host_pipeline_asan"

echo "== libvalignhost.so, valign-bench, libcpuref.so with $SAN"
g++ -std=c++14 $SAN -fPIC -shared -Wall -pthread -I"$R/include" "$CS/valign_host.cpp" -o "$OUT/libvalignhost.so" -ldl
g++ -std=c++14 $SAN -Wall -I"$R/include" "$CS/valign_bench.cpp" -o "$OUT/valign-bench" -L"$OUT" -lvalignhost -Wl,-rpath,'$ORIGIN' -ldl -pthread
gcc $SAN -fopenmp -fPIC -shared -Wall "$R/oracle/cpu_ref.c" -o "$OUT/libcpuref.so"

echo "== CPU test-suite against the sanitized libraries"
cd "$R"
# python itself is not instrumented: preload the runtimes; leaks are python's own business (detect_leaks=0),
# everything else aborts the test that triggered it
VALIGN_SANITIZED_DIR="$OUT" \
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" \
ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:allocator_may_return_null=1" \
UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1" \
python -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@"
echo "sanitize: all green"
