#!/usr/bin/env python3
"""Host threads vs wall time of the host-pointer path (1,048,576 pairs of 150 x 500, affine scoring): the container has a
CPU quota (16 CPUs on the pool), the HIP runtime's own threads and the copy issuer need some of it."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from versalignlib_amd import hipkernel, synth      # noqa: E402
from tools.hostpath_sweep import engine, best       # noqa: E402

R, F, n, blk = 150, 500, 1 << 20, 65536
r0, f0 = synth.make_pairs(blk, R, F, seed=3)
reads, refs = np.tile(r0, (n // blk, 1)), np.tile(f0, (n // blk, 1))
rows = np.zeros((n, 2, R + F), dtype=np.uint8)
idx = np.zeros((n, 4), dtype=np.int16)
hipkernel.host_register(rows)
hipkernel.host_register(idx)
for mb in (128, 256):
    for th in (6, 8, 10, 12, 14, 16):
        eng = engine({"VALIGN_HIP_ALIGN_CHUNK_BYTES": mb << 20}, R, F)
        ms = best(lambda: eng.align_host(0, reads, refs, threads=th, out=(rows, idx)), 6)
        d = eng.describe(0, n)
        print("align flat registered, chunk %3d MB, %2d threads: %6.2f ms   gather %.2f wait %.2f" % (mb, th, ms, d["host_gather_ms"], d["host_wait_ms"]), flush=True)
        eng.close()
hipkernel.host_unregister(rows)
hipkernel.host_unregister(idx)
for mb in (32, 48):
    for th in (6, 8, 10, 12, 14, 16):
        eng = engine({"VALIGN_HIP_CHUNK_BYTES": mb << 20}, R, F)
        ms = best(lambda: eng.score_host(0, reads, refs, threads=th), 6)
        d = eng.describe(0, n)
        print("score (4-bit classes), chunk %2d MB, %2d threads: %6.2f ms   gather %.2f wait %.2f" % (mb, th, ms, d["host_gather_ms"], d["host_wait_ms"]), flush=True)
        eng.close()
