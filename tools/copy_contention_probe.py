#!/usr/bin/env python3
"""Does a D2H copy running beside the alignment kernels slow them down?  (On this ROCm the device-to-host copy of
hipMemcpyAsync is a blit KERNEL, __amd_rocclr_copyBuffer, not an SDMA transfer -- rocprofv3 kernel trace of
tools/align_pipeline_trace.py.)  16 align_device calls of 65,536 pairs in stream order, alone and with 1.36 GB going
device -> pinned host / 0.68 GB host -> device on other streams meanwhile."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                        # noqa: E402
from versalignlib_amd import hipkernel              # noqa: E402

R, F, n, per = 150, 500, 1 << 20, 1 << 16
dev = torch.device("cuda", 0)
reads, refs = bench.synth_on_device(n, dev, seed=5)
eng = hipkernel.Engine(R, F, hipkernel.Scoring.make(2, -1, -3, -3, -5, -1, -5, -1))
rows = torch.empty((per, 2, R + F), dtype=torch.uint8, device=dev)
idx = torch.empty((per, 4), dtype=torch.int16, device=dev)
big_dev = torch.empty(1363148800, dtype=torch.uint8, device=dev)
big_host = torch.empty(1363148800, dtype=torch.uint8).pin_memory()
in_host = torch.empty(681574400, dtype=torch.uint8).pin_memory()
in_dev = torch.empty(681574400, dtype=torch.uint8, device=dev)
s_k, s_out, s_in = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()


def kernels():
    with torch.cuda.stream(s_k):
        for r in range(n // per):
            eng.align_device(0, reads[r * per:(r + 1) * per], refs[r * per:(r + 1) * per], rows, idx, stream=s_k)


def run(d2h, h2d, label):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if d2h:
            with torch.cuda.stream(s_out):
                big_host.copy_(big_dev, non_blocking=True)
        if h2d:
            with torch.cuda.stream(s_in):
                in_dev.copy_(in_host, non_blocking=True)
        kernels()
        s_k.synchronize()
        k_ms = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
        all_ms = (time.perf_counter() - t0) * 1e3
        best = min(best, k_ms)
    print("%-46s kernels done after %6.2f ms (everything after %6.2f ms)" % (label, best, all_ms), flush=True)


kernels()
run(False, False, "16 x 65,536 pairs, kernels alone")
run(True, False, "... with a 1.36 GB D2H copy beside them")
run(False, True, "... with a 0.68 GB H2D copy beside them")
run(True, True, "... with both")
eng.close()
