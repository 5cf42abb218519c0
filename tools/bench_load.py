#!/usr/bin/env python
"""Map-loading wall time of the CLI at the headline size (60 views @ 1920x1080), three ways:
  (a) the reference's way: float64 host decode of the 8-bit files + float constructor,
  (b) pixel codes uploaded and decoded on the GPU (PMVO.from_u8; PMVO.py's default),
  (c) the same from one memory-mapped maps pack (monohair_amd/mapspack.py).
    python tools/bench_load.py [--views 60] [--size 1920 1080]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monohair_amd import mapspack, synth  # noqa: E402
from monohair_amd import pmvo_utils as U  # noqa: E402
from monohair_amd.camera import load_cam, parsing_camera  # noqa: E402
from monohair_amd.pmvo import PMVO  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=60)
ap.add_argument("--size", type=int, nargs=2, default=[1920, 1080])
a = ap.parse_args()
H, W = a.size
tmp = tempfile.mkdtemp(prefix="mhload_")
t0 = time.time()
base = synth.write_case(tmp, "case", V=a.views, H=H, W=W, res=32)
t_write = time.time() - t0
camera = parsing_camera(load_cam(os.path.join(base, "ours/cam_params.json")), os.path.join(base, "capture_images"))
p = lambda d: os.path.join(base, d)   # noqa: E731
kw = dict(device="cuda:0", image_size=[H, W], patch_size=7, conf_threshold=0.15)
torch.zeros(1, device="cuda:0")


def timed(fn):
    torch.cuda.synchronize()
    t = time.time()
    pm = fn()
    torch.cuda.synchronize()
    dt = time.time() - t
    del pm
    return round(dt, 3)


def float_way():
    Ori, Conf = U.Load_Ori_And_Conf(camera, p("best_ori"), p("conf"))
    return PMVO(camera, U.load_depth(camera, p("render_depth")), Ori, Conf, U.load_mask(camera, p("hair_mask")), **kw)


def u8_way():
    o, c, m = U.load_maps_u8(camera, p("best_ori"), p("conf"), p("hair_mask"))
    return PMVO.from_u8(camera, U.load_depth_plane(camera, p("render_depth")), o, c, m, **kw)


def pack_way():
    m = mapspack.read_pack(p("maps.mhpk"), views=list(camera.keys()))
    return PMVO.from_u8(camera, m["depth"], m["ori"], m["conf"], m["mask"], **kw)


res = {"views": a.views, "image": [H, W], "write_case_s": round(t_write, 1)}
res["float64_host_decode_s"] = timed(float_way)
res["u8_gpu_decode_s"] = timed(u8_way)
t = time.time()
mapspack.pack_case(camera, p("best_ori"), p("conf"), p("hair_mask"), p("render_depth"), p("maps.mhpk"))
res["pack_write_s"] = round(time.time() - t, 3)
res["pack_bytes"] = os.path.getsize(p("maps.mhpk"))
res["maps_pack_s"] = timed(pack_way)
res["maps_pack_again_s"] = timed(pack_way)
print(json.dumps(res))
