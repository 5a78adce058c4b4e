"""Time the score kernel for several forced geometries on one GPU (developer tool)."""
import argparse
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from versalignlib_amd import hipkernel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=262144)
    ap.add_argument("--R", type=int, default=150)
    ap.add_argument("--F", type=int, default=500)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--geoms", default="0x0,8x20,16x10,16x12,32x8,64x12")
    ap.add_argument("--affine", type=int, default=0, help="1: open -5 / extend -1 both ways; 2: four different scores")
    ap.add_argument("--opt", type=int, default=0)
    ap.add_argument("--band", type=int, default=0)
    ap.add_argument("--width", type=int, default=0, help="score_width: 0 auto, 16, 32 (int32 cells, strip path)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    refs = lut[torch.randint(0, 4, (a.n, a.F), device=dev, generator=g)]
    off = torch.randint(0, max(a.F - a.R, 0) + 1, (a.n, 1), device=dev, generator=g)
    cols = (off + torch.arange(a.R, device=dev)[None, :]).clamp(max=a.F - 1)
    reads = torch.gather(refs, 1, cols)
    sub = torch.rand((a.n, a.R), device=dev, generator=g) < 0.15
    reads = torch.where(sub, lut[torch.randint(0, 4, (a.n, a.R), device=dev, generator=g)], reads).contiguous()
    sc = hipkernel.Scoring.make()
    if a.affine == 1:
        sc = hipkernel.Scoring.make(2, -1, -3, -3, -5, -1, -5, -1)
    elif a.affine == 2:
        sc = hipkernel.Scoring.make(2, -1, -3, -3, -5, -1, -4, -2)
    for geom in a.geoms.split(","):
        G, K = (int(x) for x in geom.split("x"))
        try:
            eng = hipkernel.Engine(a.R, a.F, sc, group_lanes=G, rows_per_lane=K)
        except hipkernel.HipKernelError as e:
            print(geom, "unavailable:", e)
            continue
        if a.band:
            eng.set_band_width(a.band)
        if a.width:
            eng.set_score_width(a.width)
        out = eng.score_device(a.opt, reads, refs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            eng.score_device(a.opt, reads, refs, out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        d = eng.describe(a.opt, a.n)
        print(json.dumps({"geom": geom, "ms": round(ms, 3), "gcups": round(a.n * a.R * a.F / ms / 1e6, 1),
                          "checksum": int(out.to(torch.int64).sum().item()), **{k: d[k] for k in ("group_lanes", "rows_per_lane", "lds_per_wave", "waves_per_block")}}))
        eng.close()


if __name__ == "__main__":
    main()
