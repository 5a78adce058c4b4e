"""Device memory before and after 40 device-resident calls of mixed modes and batch sizes on two engines (scratch, helper
stream and events are per engine and persistent: the free memory must not move), and after closing them
(developer tool; run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from versalignlib_amd import hipkernel
dev = torch.device("cuda:0")
n = 1 << 19
reads, refs = bench.synth_on_device(n, dev, seed=7)
eng = hipkernel.Engine(150, 500, hipkernel.Scoring.make(2, -1, -3, -3, **bench.AFFINE))
lin = hipkernel.Engine(150, 500, hipkernel.Scoring.make(2, -1, -3, -3))
rows = torch.empty((n, 2, 650), dtype=torch.uint8, device=dev); idx = torch.empty((n, 4), dtype=torch.int16, device=dev)
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
for e in (eng, lin):
    for opt in (0, 1):
        e.align_device(opt, reads, refs, rows, idx); e.score_device(opt, reads, refs)
f0 = free()
for it in range(40):
    e = (eng, lin)[it & 1]
    m = n if it % 3 else n // 3 + 17
    e.align_device(it % 2, reads[:m], refs[:m], rows[:m], idx[:m])
    e.score_device((it + 1) % 2, reads[:m], refs[:m])
f1 = free()
print("free before %.1f MB, after %.1f MB, delta %.1f MB" % (f0 / 2**20, f1 / 2**20, (f0 - f1) / 2**20))
# length-sorted batching on the device: its contexts (packed copies, bins, places) are per engine and persistent too
eng.set_ragged_batching(2)
eng.score_device(0, reads, refs); eng.score_device(1, reads[: n // 2], refs[: n // 2])
f_r0 = free()
for it in range(10):
    eng.score_device(it % 2, reads[: n - 1000 * it], refs[: n - 1000 * it])
f_r1 = free()
print("length-sorted calls: free before %.1f MB, after %.1f MB, delta %.1f MB" % (f_r0 / 2**20, f_r1 / 2**20, (f_r0 - f_r1) / 2**20))
eng.set_ragged_batching(0)
eng.close(); lin.close()
print("after close free %.1f MB" % (free() / 2**20))
