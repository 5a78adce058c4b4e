"""Target for `rocprofv3 --kernel-trace --memory-copy-trace`: three score_host calls of 1 M pairs through the chunk
pipeline (developer tool; the trace shows whether the H2D copies run at PCIe speed while the host threads gather)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from versalignlib_amd import hipkernel, synth

R, F, blk, n = 150, 500, 4096, 1 << 20
r0, f0 = synth.make_pairs(blk, R, F, seed=3)
reads = np.ascontiguousarray(np.tile(r0, (n // blk, 1)))
refs = np.ascontiguousarray(np.tile(f0, (n // blk, 1)))
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
eng = hipkernel.Engine(R, F)
eng.score_host(0, reads, refs, threads=threads)
for _ in range(3):
    t0 = time.perf_counter()
    eng.score_host(0, reads, refs, threads=threads)
    d = eng.describe(0, n)
    print("call %.2f ms gather %.2f wait %.2f drain %.2f" % ((time.perf_counter() - t0) * 1e3, d["host_gather_ms"], d["host_wait_ms"], d["host_drain_ms"]))
eng.close()
