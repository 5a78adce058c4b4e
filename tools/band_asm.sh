#!/bin/bash
# Device assembly of engine_long.hip (band and long-read kernels) -> /tmp/valign_long_unit.s, then one score_band_kernel
# instance (default: <16, shared gap, unit delay, linear>) -> /tmp/band_kernel.s
cd "$(dirname "$0")/../versalignlib_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 --cuda-device-only -S -I../../include -I. engine_long.hip -o /tmp/valign_long_unit.s 2>/tmp/band_asm.err || { tail -20 /tmp/band_asm.err; exit 1; }
k=${1:-_ZN6valign17score_band_kernelILi16ELb1ELb1ELb0EEEvNS_8BandArgsE}
awk -v k="^$k:" '$0 ~ k {f=1} f&&/s_endpgm/{exit} f' /tmp/valign_long_unit.s > /tmp/band_kernel.s
wc -l /tmp/band_kernel.s
