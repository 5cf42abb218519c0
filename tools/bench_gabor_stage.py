#!/usr/bin/env python
"""Wall time of the whole Gabor stage (GaborFilter.batch_generate: read capture images, DoG, bank, write best_ori/,
conf/ and Ori/) on V synthetic 1920x1080 views, PNG or JPG like the capture.
    python tools/bench_gabor_stage.py [--views 60] [--ext png]
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monohair_amd.gabor import batch_generate  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=60)
ap.add_argument("--ext", default="png")
ap.add_argument("--threads", type=int, default=8)
a = ap.parse_args()
root = tempfile.mkdtemp(prefix="mhgabor_")
os.makedirs(os.path.join(root, "capture_images"))
rng = np.random.default_rng(0)
r, c = np.meshgrid(np.arange(1920), np.arange(1080), indexing="ij")
for v in range(a.views):
    th = 0.1 * v
    im = (127 + 60 * np.cos(2 * np.pi * (r * np.cos(th) + c * np.sin(th)) / 4.0) + rng.normal(0, 5, r.shape)).clip(0, 255)
    Image.fromarray(np.repeat(im.astype(np.uint8)[..., None], 3, 2)).save(
        os.path.join(root, "capture_images", "%03d.%s" % (v, a.ext)))
import torch  # noqa: E402

torch.zeros(1, device="cuda:0")
batch_generate(root, "capture_images", io_threads=a.threads)      # warm: code objects, file cache
t0 = time.perf_counter()
batch_generate(root, "capture_images", io_threads=a.threads)
dt = time.perf_counter() - t0
print({"views": a.views, "ext": a.ext, "io_threads": a.threads, "stage_s": round(dt, 2), "views_per_s": round(a.views / dt, 1)})
