#!/usr/bin/env python
"""How long does the HOST need to enqueue one forward() (python + ctypes + allocator), against the GPU time per iteration?
In the 8-bit regime an iteration is ~0.27 ms of GPU work: if the enqueue takes as long, the loop is host-bound."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monohair_amd import synth  # noqa: E402
from monohair_amd.camera import cameras_from_list  # noqa: E402
from monohair_amd.pmvo import PMVO  # noqa: E402

dev = torch.device("cuda:0")
V, H, W = 60, 1920, 1080
sc = synth.make_scene_codes(V, H, W, device=dev, seed=0)
cams = cameras_from_list(sc["cams"])
pm = PMVO.from_u8(cams, sc["depth"], sc["ori_u8"], sc["conf_u8"], sc["mask_u8"], device=dev, image_size=[H, W], patch_size=7,
                  visible_threshold=1, conf_threshold=0.15)
cand = synth.candidate_points(res=256, seed=0)
surf = np.concatenate([pm.filter_points(cand[i:i + 200000])[0].cpu().numpy() for i in range(0, len(cand), 200000)])
pts = cand[surf]
chunks = [pts[i * 5000:(i + 1) * 5000] for i in range(len(pts) // 5000)]
for ns in (1, 3):
    streams = pm.side_streams(ns)

    def step(i):
        with torch.cuda.stream(streams[i % ns]):
            return pm.forward(chunks[i % len(chunks)])

    for i in range(60):
        step(i)
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for i in range(K):
        step(i)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("streams %d: enqueue %.1f us / iteration, complete %.1f us / iteration (%.0f it/s)" % (
        ns, t_enq / K * 1e6, t_all / K * 1e6, K / t_all))
import cProfile
import pstats

pr = cProfile.Profile()
pr.enable()
for i in range(200):
    step(i)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
