"""Large gap / match scores through every cell format the engine can pick (half floats, int16, int32 strips, equality-test
fallbacks), scores and alignments against the oracle (developer tool; run on the GPU box)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpu_ref
from versalignlib_amd import hipkernel, synth
R, F, n = 150, 500, 300
reads, refs = synth.make_pairs(n, R, F, seed=5, indel_rate=0.02, n_run_frac=0.05, short_frac=0.1)
dr, df = torch.from_numpy(reads).cuda(), torch.from_numpy(refs).cuda()
bad = 0
for name, args, kw in [
    ("gap-60", (2, -1, -60, -60), {}),
    ("gap-60/-5", (2, -1, -60, -5), {}),
    ("gap-200", (5, -4, -200, -200), {}),
    ("aff ext-40", (2, -1, -3, -3), dict(open_read=-90, ext_read=-40, open_ref=-90, ext_ref=-40)),
    ("aff ext-40/-1", (2, -1, -3, -3), dict(open_read=-90, ext_read=-40, open_ref=-9, ext_ref=-1)),
    ("aff ext-300", (9, -9, -3, -3), dict(open_read=-400, ext_read=-300, open_ref=-400, ext_ref=-300)),
    ("match100", (100, -90, -50, -70), {}),
]:
    osc = cpu_ref.Scoring.make(*args, **kw); hsc = hipkernel.Scoring.make(*args, **kw)
    aff = bool(kw)
    try:
        eng = hipkernel.Engine(R, F, hsc)
    except Exception as e:
        print(name, "engine refused:", e); continue
    for opt in (0, 1):
        try:
            got = eng.score_device(opt, dr, df).cpu().numpy()
            exp = cpu_ref.score(opt, reads, refs, osc, threads=8, affine=aff)
            ok = np.array_equal(got, exp)
            cells = eng.describe(opt)["score_cells"]
        except Exception as e:
            ok, cells = "raised: %s" % e, "-"
        try:
            rows, idx = eng.align_device(opt, dr, df)
            erows, eidx = cpu_ref.align(opt, reads, refs, osc, threads=8, affine=aff)
            aok = bool(np.array_equal(idx.cpu().numpy(), eidx) and np.array_equal(rows.cpu().numpy(), erows))
        except Exception as e:
            aok = "raised: %s" % str(e)[:80]
        print(name, "opt", opt, "score", ok, cells, "align", aok)
        if ok is False or aok is False: bad += 1
    eng.close()
print("BAD", bad)
