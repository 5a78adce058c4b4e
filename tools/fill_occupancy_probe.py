#!/usr/bin/env python3
"""Would a third wave per SIMD help the alignment fill kernels?  BASELINE config 3's kernel (NW, symmetric affine, tagged
cells, 16 x 10) at reference lengths whose LDS footprint lets two or three 4-wave blocks share a CU: time per wave-step
(F + 15 steps of 8 pairs).  The fill's instructions are half full-rate ANDs / bit operations, unlike the score kernels'.
Run on the GPU box."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from versalignlib_amd import hipkernel, synth        # noqa: E402

R = 150
sc = hipkernel.Scoring.make(2, -1, -3, -3, -5, -1, -5, -1)
for F in (56, 64, 72, 100, 128, 200, 500):
    n = max(8192, int((1 << 20) * 500 / F) // 8192 * 8192)
    n = min(n, 4 << 20)
    reads, refs = synth.make_pairs(8192, R, F, seed=F)
    d_reads = torch.from_numpy(reads).cuda().repeat(n // 8192, 1).contiguous()
    d_refs = torch.from_numpy(refs).cuda().repeat(n // 8192, 1).contiguous()
    eng = hipkernel.Engine(R, F, sc)
    d = eng.describe(1, n)
    rows = torch.empty((n, 2, R + F), dtype=torch.uint8, device="cuda")
    idx = torch.empty((n, 4), dtype=torch.int16, device="cuda")
    eng.align_device(1, d_reads, d_refs, rows, idx)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.align_device(1, d_reads, d_refs, rows, idx)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    wave_steps = (n / 8) * (F + 15)
    blocks_per_cu = (160 * 1024) // (d["lds_per_wave"] * d["waves_per_block"])
    print(json.dumps({"F": F, "pairs": n, "ms": round(best, 3), "ns_per_wave_step_per_cu_slot": round(best * 1e6 / wave_steps * 256, 2),
                      "lds_per_wave": d["lds_per_wave"], "waves_per_cu": blocks_per_cu * d["waves_per_block"],
                      "geometry": "%dx%d" % (d["group_lanes"], d["rows_per_lane"]), "gcups": round(n * R * F / best / 1e6, 1)}))
    eng.close()
