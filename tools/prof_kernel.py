"""Launch the score kernels a few times on the bench batch (target for rocprofv3 runs)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from versalignlib_amd import hipkernel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1 << 20)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--modes", default="sw_affine16,sw_affine,sw_linear,nw_linear",
                    help="<alg>_<gap model>; a trailing 16 forces the int16-cell kernels (bench.py's headline)")
    ap.add_argument("--geom", default="0x0")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    reads, refs = bench.synth_on_device(a.pairs, dev, seed=2000)
    G, K = (int(x) for x in a.geom.split("x"))
    out = torch.empty(a.pairs, dtype=torch.int16, device=dev)
    for mode in a.modes.split(","):
        alg, gap = mode.split("_")
        int16_cells = gap.endswith("16")
        gap = gap[:-2] if int16_cells else gap
        sc = hipkernel.Scoring.make(2, -1, -3, -3, **(bench.AFFINE if gap == "affine" else {}))
        eng = hipkernel.Engine(bench.R, bench.F, sc, group_lanes=G, rows_per_lane=K)
        if int16_cells:
            eng.set_half_float_cells(0)
        for _ in range(a.reps):
            eng.score_device(0 if alg == "sw" else 1, reads, refs, out)
        torch.cuda.synchronize()
        print(mode, eng.describe(0, a.pairs), int(out.to(torch.int64).sum().item()))
        eng.close()


if __name__ == "__main__":
    main()
