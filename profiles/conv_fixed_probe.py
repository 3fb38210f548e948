"""Fixed cost of the level-0 3x3 convolutions (B2 128^2 -> Cout 320, the step's second-largest class): time against the number of K-tiles
(Cin = 64 .. 960), hipGraph-timed so that the host launch cost is out of the picture; the intercept of the line is what a launch costs besides its K loop."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldx_amd as ldx
from profiles.kprobe import L, p, st, timeit_graph

B, H, Cout = 2, 128, 320
rows = []
for Cin in (64, 128, 192, 256, 320, 640, 960):
    X = torch.randn(B, H, H, Cin, device="cuda").bfloat16()
    W = (torch.randn(Cout, 9 * Cin, device="cuda") / (3 * Cin ** 0.5)).bfloat16()
    Y = torch.empty(B * H * H, Cout, device="cuda", dtype=torch.bfloat16)
    R = torch.randn(B * H * H, Cout, device="cuda").bfloat16()
    bias = torch.randn(Cout, device="cuda")
    for tag, res in (("plain", None), ("bias+residual", R)):
        fn = lambda: L.ldx_op_conv3x3(p(X), Cin, p(W), B, H, H, Cin, Cout, 1, H, H, 0, p(bias) if res is not None else None, None, 0, p(res), Cout if res is not None else 0, p(Y), Cout, 0, st())
        assert fn() == 0
        us = timeit_graph(fn, reps=20) * 1e3
        kt = 9 * Cin // 64
        rows.append((Cin, tag, kt, us))
        print(f"conv 128^2 {Cin:4d}->{Cout} {tag:14s} K-tiles {kt:4d}  {us:7.1f} us  {2.0 * B * H * H * Cout * 9 * Cin / us / 1e6:7.1f} TFLOP/s", flush=True)
for tag in ("plain", "bias+residual"):
    pts = [(k, u) for (_, t, k, u) in rows if t == tag]
    n = len(pts); sx = sum(k for k, _ in pts); sy = sum(u for _, u in pts); sxx = sum(k * k for k, _ in pts); sxy = sum(k * u for k, u in pts)
    slope = (n * sxy - sx * sy) / (n * sxx - sx * sx); icpt = (sy - slope * sx) / n
    print(f"{tag}: {slope:.3f} us per K-tile, intercept {icpt:.1f} us")
