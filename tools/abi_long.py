import sys, time, numpy as np
sys.path.insert(0, '.')
from versalignlib_amd import build, host, synth
R=F=10000; n=8192
reads, refs = synth.make_pairs(n, R, F, seed=5, sub_rate=0.1)
for keys in (dict(band_width=512, score_width=32), dict(band_width=512, score_width=32, host_packing=0), dict()):
    with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=16, **keys) as hip:
        hip.score_alignments(0, reads, refs)
        t=[]
        for _ in range(3):
            t0=time.perf_counter(); s=hip.score_alignments(0, reads, refs); t.append(time.perf_counter()-t0)
        print(keys, "score_alignments: %.1f ms per %d pairs (min of 3), checksum %d" % (min(t)*1e3, n, int(s.astype(np.int64).sum())), flush=True)
n2=1024
with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=16) as hip:
    hip.compute_alignments(0, reads[:n2], refs[:n2], normalise=False)
    t0=time.perf_counter(); hip.compute_alignments(0, reads[:n2], refs[:n2], normalise=False); print("compute_alignments SW: %.1f ms per %d pairs" % ((time.perf_counter()-t0)*1e3, n2))
