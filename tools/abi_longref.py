"""score_alignments through the plugin ABI for short reads against a long reference (developer tool)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from versalignlib_amd import build, host, synth
R, F, n = 150, 8000, 65536
reads, refs = synth.make_pairs(n, R, F, seed=9, sub_rate=0.1)
for keys in (dict(), dict(score_gap_open_read=-5, score_gap_extend_read=-1, score_gap_open_ref=-5, score_gap_extend_ref=-1)):
    with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=16, **keys) as hip:
        hip.score_alignments(0, reads, refs)
        t = []
        for _ in range(3):
            t0 = time.perf_counter(); s = hip.score_alignments(0, reads, refs); t.append(time.perf_counter() - t0)
        print("affine" if keys else "linear", "score_alignments: %.1f ms per %d pairs of %d x %d (min of 3)" % (min(t) * 1e3, n, R, F), flush=True)
        print(hip.drain_log().strip().splitlines()[-1][-330:])
