import sys, time, numpy as np
sys.path.insert(0, '.')
from versalignlib_amd import build, host, synth
R=F=10000; n=4096
reads, refs = synth.make_pairs(n, R, F, seed=5, sub_rate=0.1)
with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=16) as hip:
    hip.compute_alignments(0, reads[:256], refs[:256], normalise=False)
    for m in (4096, 4096, 1024):
        t0=time.perf_counter(); hip.compute_alignments(0, reads[:m], refs[:m], normalise=False); print("compute_alignments SW: %.1f ms per %d pairs" % ((time.perf_counter()-t0)*1e3, m), flush=True)
    print(hip.drain_log()[-600:])
