"""Time compute_alignments' device path (fill + traceback) on the bench batch (developer tool)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from versalignlib_amd import hipkernel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1 << 20)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--geoms", default="0x0")
    ap.add_argument("--policy", type=int, default=0, help="1: SSE/AVX tie-breaks")
    ap.add_argument("--models", default="linear,affine", help="gap models to time (affine: open -5, extend -1; BASELINE config 3)")
    ap.add_argument("--R", type=int, default=bench.R)
    ap.add_argument("--F", type=int, default=bench.F)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    bench.R, bench.F = a.R, a.F                     # (other shapes: reads beyond 2048 rows take the row strips)
    reads, refs = bench.synth_on_device(a.pairs, dev, seed=2000, R=a.R, F=a.F)
    AL = bench.R + bench.F
    rows = torch.empty((a.pairs, 2, AL), dtype=torch.uint8, device=dev)
    idx = torch.empty((a.pairs, 4), dtype=torch.int16, device=dev)
    for geom, model in ((g, m) for g in a.geoms.split(",") for m in a.models.split(",")):
        G, K = (int(x) for x in geom.split("x"))
        if model == "affine" and a.policy:
            continue
        sc = hipkernel.Scoring.make(2, -1, -3, -3, **(bench.AFFINE if model == "affine" else {}))
        eng = hipkernel.Engine(bench.R, bench.F, sc, group_lanes=G, rows_per_lane=K)
        if a.policy:
            eng.set_traceback_policy(a.policy)
        for opt, name in ((0, "sw_" + model), (1, "nw_" + model)):
            eng.align_device(opt, reads, refs, rows, idx)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                eng.align_device(opt, reads, refs, rows, idx)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            d = eng.describe(opt, a.pairs)
            print(json.dumps({"mode": name + "_align", "geom": "%dx%d" % (d["group_lanes"], d["rows_per_lane"]),
                              "ms": round(ms, 3), "gcups": round(a.pairs * bench.R * bench.F / ms / 1e6, 1),
                              "start_checksum": int(idx[:, 0].to(torch.int64).sum().item())}))
        eng.close()


if __name__ == "__main__":
    main()
