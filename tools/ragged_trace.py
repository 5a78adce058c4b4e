"""Kernel timeline of the device-side length-sorted batching (run under rocprofv3 --kernel-trace on the GPU box):
1 M pairs of 150 x 500, a mixed-length batch and a full-length one, ragged_batching = 2, three calls each."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from versalignlib_amd import hipkernel, synth
R, F, n, blk = 150, 500, 1 << 20, 1 << 16
AFF = dict(open_read=-5, ext_read=-1, open_ref=-5, ext_ref=-1)
for ragged in (True, False):
    r0, f0 = (synth.make_ragged_pairs if ragged else synth.make_pairs)(blk, R, F, seed=3)
    reads, refs = np.tile(r0, (n // blk, 1)), np.tile(f0, (n // blk, 1))
    d_reads, d_refs = torch.from_numpy(reads).cuda(), torch.from_numpy(refs).cuda()
    eng = hipkernel.Engine(R, F, hipkernel.Scoring.make(2, -1, -3, -3, **AFF))
    out = torch.empty(n, dtype=torch.int16, device="cuda")
    eng.set_ragged_batching(2)
    for _ in range(3):
        eng.score_device(0, d_reads, d_refs, out)
        torch.cuda.synchronize()
    eng.close()
