#!/usr/bin/env python3
"""Length-sorted batching on the device (ragged_kernels.hip.h): 1,048,576 pairs of 150 x 500 whose reads and references
keep a uniformly drawn prefix (10-100 %) and are NUL-padded, as a FASTA of mixed lengths looks at the plugin boundary.
Device-resident batch (valign_hip_score_device, wall time of the call including its wait for the histogram) and the plugin
ABI with scattered host pointers, ragged_batching 0 / 1 / 2; SW and the NW variant, linear and affine gaps.  GCUPS are counted
on the padded shape, as the reference does.  Run on the GPU box."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from versalignlib_amd import build, hipkernel, host, synth      # noqa: E402

R, F, n, blk = 150, 500, 1 << 20, 1 << 16
AFF = dict(open_read=-5, ext_read=-1, open_ref=-5, ext_ref=-1)


def batch(ragged):
    r0, f0 = (synth.make_ragged_pairs if ragged else synth.make_pairs)(blk, R, F, seed=3)
    return np.tile(r0, (n // blk, 1)), np.tile(f0, (n // blk, 1))


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    for ragged_input in (True, False):
        reads, refs = batch(ragged_input)
        d_reads, d_refs = torch.from_numpy(reads).cuda(), torch.from_numpy(refs).cuda()
        for affine in (True, False):
            sc = hipkernel.Scoring.make(2, -1, -3, -3, **AFF) if affine else hipkernel.Scoring.make()
            eng = hipkernel.Engine(R, F, sc)
            out = torch.empty(n, dtype=torch.int16, device="cuda")
            ref_scores = None
            for alg in (0, 1):
                for mode in (0, 2, 1):
                    eng.set_ragged_batching(mode)
                    eng.score_device(alg, d_reads, d_refs, out)
                    torch.cuda.synchronize()
                    best = 1e9
                    for _ in range(5):
                        t0 = time.perf_counter()
                        eng.score_device(alg, d_reads, d_refs, out)
                        torch.cuda.synchronize()
                        best = min(best, time.perf_counter() - t0)
                    d = eng.describe(alg, n)
                    if mode == 0:
                        ref_scores = out.clone()
                    same = bool(torch.equal(out, ref_scores))
                    print(json.dumps({"batch": "ragged" if ragged_input else "uniform", "call": "score_device", "alg": "SW" if alg == 0 else "NW",
                                      "gaps": "affine" if affine else "linear", "ragged_batching": mode, "ms": round(best * 1e3, 3),
                                      "padded_gcups": round(synth.gcups(n, R, F, best), 1), "launches": d["ragged_launches"],
                                      "cell_fraction": d["ragged_cell_fraction"], "same_scores_as_mode_0": same}), flush=True)
            eng.close()
        del d_reads, d_refs
        keys = dict(score_gap_open_read=-5, score_gap_extend_read=-1, score_gap_open_ref=-5, score_gap_extend_ref=-1)
        base = None
        for mode in (0, 1, 2):
            with host.Plugin(build.HIP_PLUGIN, R, F, num_threads=threads, ragged_batching=mode, **keys) as k:
                k.score_alignments(0, reads, refs, scattered=True)
                runs = [k.score_alignments(0, reads, refs, scattered=True) for _ in range(5)]
                best = min(r[1] for r in runs)
                if mode == 0:
                    base = runs[0][0].copy()
                phases = [ln for ln in k.drain_log().splitlines() if "score done" in ln]
                print(json.dumps({"batch": "ragged" if ragged_input else "uniform", "call": "score_alignments(SW, affine) via the ABI", "threads": threads,
                                  "ragged_batching": mode, "ms": round(best * 1e3, 2), "padded_gcups_pcie_inclusive": round(synth.gcups(n, R, F, best), 1),
                                  "same_scores_as_mode_0": bool(np.array_equal(runs[-1][0], base)),
                                  "host_phases": json.loads(phases[-1].split("host phases ")[-1]) if phases else None}), flush=True)


if __name__ == "__main__":
    main()
