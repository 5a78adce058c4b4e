#!/usr/bin/env python3
"""score_band_kernel, 10 kbp x 10 kbp at 512 diagonals: kernel time against the number of waves per CU in the launch
(4 pairs per wave, 256 CUs) -- the step pattern shows how many waves a CU really runs side by side."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                        # noqa: E402
from versalignlib_amd import hipkernel              # noqa: E402

R = F = 10000
dev = torch.device("cuda", 0)
reads, refs = bench.synth_on_device(1024 * 12, dev, seed=5, R=R, F=F)
eng = hipkernel.Engine(R, F)
eng.set_band_width(512)
eng.set_score_width(32)
out = torch.empty(1024 * 12, dtype=torch.int16, device=dev)
eng.score_device(0, reads[:1024], refs[:1024], out[:1024])
torch.cuda.synchronize()
for k in range(1, 13):
    n = 1024 * k
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.score_device(0, reads[:n], refs[:n], out[:n])
    e1.record()
    torch.cuda.synchronize()
    print("%2d waves per CU (%5d pairs): %7.3f ms" % (k, n, e0.elapsed_time(e1)), flush=True)
eng.close()
