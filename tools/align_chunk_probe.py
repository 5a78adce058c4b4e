#!/usr/bin/env python3
"""Device-side cost of compute_alignments(SW, affine scoring) as a function of the batch size one call gets: what a
chunk of the host-pointer pipeline costs on the device (fill + walk in stream order), ms per million pairs."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                        # noqa: E402
from versalignlib_amd import hipkernel              # noqa: E402

R, F = 150, 500
dev = torch.device("cuda", 0)
reads, refs = bench.synth_on_device(1 << 20, dev, seed=5)
for name, sc in (("affine", hipkernel.Scoring.make(2, -1, -3, -3, -5, -1, -5, -1)), ("linear", hipkernel.Scoring.make())):
    eng = hipkernel.Engine(R, F, sc)
    for n in (1 << 20, 1 << 18, 1 << 17, 68000, 1 << 15):
        rows = torch.empty((n, 2, R + F), dtype=torch.uint8, device=dev)
        idx = torch.empty((n, 4), dtype=torch.int16, device=dev)
        reps = (1 << 20) // n
        eng.align_device(0, reads[:n], refs[:n], rows, idx)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(reps):
            eng.align_device(0, reads[r * n:(r + 1) * n], refs[r * n:(r + 1) * n], rows, idx)
        e1.record()
        torch.cuda.synchronize()
        print("%s SW alignments, %7d pairs per call x %2d calls in stream order: %.2f ms per %d pairs" %
              (name, n, reps, e0.elapsed_time(e1), n * reps), flush=True)
    eng.close()
