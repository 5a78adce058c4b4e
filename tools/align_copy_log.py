"""One valign_hip_align_host call of 200,000 pairs (three chunks) for AMD_LOG_LEVEL=4 runs: which path do the
runtime's result copies take?  (developer tool)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from versalignlib_amd import hipkernel, synth

R, F, n = 150, 500, 200000
reads, refs = synth.make_pairs(n, R, F, seed=3)
eng = hipkernel.Engine(R, F)
print("=== WARM CALL", flush=True)
eng.align_host(0, reads, refs, threads=8)
print("=== TRACED CALL", flush=True)
sys.stderr.write("=== TRACED CALL\n")
sys.stderr.flush()
eng.align_host(0, reads, refs, threads=8)
sys.stderr.write("=== END\n")
eng.close()
