"""Target for `rocprofv3 --kernel-trace --memory-copy-trace`: valign_hip_align_host (SW, affine scoring) of 1 M pairs
into registered result buffers, three calls (developer tool: the trace shows where the device idles)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from versalignlib_amd import hipkernel, synth

R, F, blk, n = 150, 500, 65536, 1 << 20
r0, f0 = synth.make_pairs(blk, R, F, seed=3)
reads = np.ascontiguousarray(np.tile(r0, (n // blk, 1)))
refs = np.ascontiguousarray(np.tile(f0, (n // blk, 1)))
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
registered = len(sys.argv) <= 2 or sys.argv[2] != "0"
eng = hipkernel.Engine(R, F, hipkernel.Scoring.make(2, -1, -3, -3, -5, -1, -5, -1))
rows = np.zeros((n, 2, R + F), dtype=np.uint8)
idx = np.zeros((n, 4), dtype=np.int16)
if registered:
    hipkernel.host_register(rows)
    hipkernel.host_register(idx)
eng.align_host(0, reads, refs, threads=threads, out=(rows, idx))
for _ in range(3):
    t0 = time.perf_counter()
    eng.align_host(0, reads, refs, threads=threads, out=(rows, idx))
    d = eng.describe(0, n)
    print("call %.2f ms gather %.2f wait %.2f drain %.2f direct_out %d" % ((time.perf_counter() - t0) * 1e3, d["host_gather_ms"], d["host_wait_ms"], d["host_drain_ms"], d["direct_out"]))
if registered:
    hipkernel.host_unregister(rows)
    hipkernel.host_unregister(idx)
eng.close()
