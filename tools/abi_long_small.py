"""Latency of small long-read calls through the plugin ABI (developer tool): 10 kbp x 10 kbp, n = 1 .. 256."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from versalignlib_amd import build, host, synth
R = F = 10000
reads, refs = synth.make_pairs(256, R, F, seed=7, sub_rate=0.1)
for name, keys in (("band 512", dict(band_width=512, score_width=32)), ("unbanded", dict()), ("unbanded affine", dict(score_gap_open_read=-5, score_gap_extend_read=-1, score_gap_open_ref=-5, score_gap_extend_ref=-1))):
    with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=8, **keys) as hip:
        for n in (1, 16, 256):
            hip.score_alignments(0, reads[:n], refs[:n])
            t = []
            for _ in range(3):
                t0 = time.perf_counter(); hip.score_alignments(0, reads[:n], refs[:n]); t.append(time.perf_counter() - t0)
            line = "%-16s n=%3d  score_alignments %.2f ms" % (name, n, min(t) * 1e3)
            if name != "band 512":
                hip.compute_alignments(0, reads[:n], refs[:n], normalise=False)
                t0 = time.perf_counter(); hip.compute_alignments(0, reads[:n], refs[:n], normalise=False)
                line += "   compute_alignments %.2f ms" % ((time.perf_counter() - t0) * 1e3)
            print(line, flush=True)
