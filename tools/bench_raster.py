#!/usr/bin/env python
"""Depth rasteriser timing: V views @ HxW of a tessellated sphere (+ bust sphere), HIP events around the calls.
    python tools/bench_raster.py [--views 60] [--size 1920 1080] [--lat 512]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from monohair_amd import synth  # noqa: E402
from monohair_amd.camera import camera_records, cameras_from_list  # noqa: E402
from monohair_amd.render import DepthRenderer  # noqa: E402
from test_raster_host import uv_sphere  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=60)
ap.add_argument("--size", type=int, nargs=2, default=[1920, 1080])
ap.add_argument("--lat", type=int, default=512)
a = ap.parse_args()
H, W = a.size
cams = synth.make_cameras(a.views, H, W, scale=1.7)
rec = camera_records(cameras_from_list(cams))
v, f = uv_sphere(synth.SPHERE_R, a.lat, 2 * a.lat)
bv, bf = uv_sphere(0.09, 64, 128)
bv = bv + np.array([0, -0.12, 0], np.float32)
r = DepthRenderer([(v, f), (bv, bf)], "cuda:0")
out = torch.empty((a.views, H, W), dtype=torch.float32, device="cuda:0")
for i in range(3):
    r.render(rec[i], H, W, 0.5, out=out[i])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(a.views):
    r.render(rec[i], H, W, 0.5, out=out[i])
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
print(json.dumps({"views": a.views, "image": [H, W], "triangles": int(len(f) + len(bf)), "vertices": int(len(v) + len(bv)),
                  "ms_total": round(ms, 2), "ms_per_view": round(ms / a.views, 3),
                  "covered_fraction": round(float((out < 255).float().mean()), 4)}))
