#!/bin/bash
# Device assembly of the main translation unit (band / long / strip kernels) -> /tmp/valign_part_main.s, then the
# maximum / subtract / add sequence of one score_band_kernel instance (default: <16, shared gap, unit delay, 2 chains>)
cd "$(dirname "$0")/../versalignlib_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 --cuda-device-only -S -I../../include -I. hip_plugin.hip -o /tmp/valign_part_main.s 2>/tmp/band_asm.err || { tail -20 /tmp/band_asm.err; exit 1; }
k=${1:-_ZN6valign17score_band_kernelILi16ELb1ELb1ELi2EEEvNS_8BandArgsE}
awk -v k="^$k:" '$0 ~ k {f=1} f&&/s_endpgm/{exit} f' /tmp/valign_part_main.s > /tmp/band_kernel.s
wc -l /tmp/band_kernel.s
