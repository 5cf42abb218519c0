"""Experiment (GPU box): throughput of the search kernel alone on two alternating streams (front end prepared once per
stream), against the full iteration -- how much of the iteration the front end costs when overlapped."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from monohair_amd import _lib, synth
from monohair_amd.camera import camera_records, cameras_from_list
from monohair_amd.pmvo import PMVO

dev = torch.device("cuda", 0)
V, H, W, N = 60, 1920, 1080, 5000
scene = synth.make_scene(V, H, W, device=dev, seed=0, quantize=False)
cams = cameras_from_list(scene["cams"])
pm = PMVO.from_planes(camera_records(cams), scene["depth"], scene["ori"], scene["conf"], scene["mask"], device=dev,
                      patch_size=7, visible_threshold=1, conf_threshold=0.15, camera=cams)
cand = synth.candidate_points(res=256, seed=0)
surf = []
for i in range(0, len(cand), 200000):
    s, _, _ = pm.filter_points(cand[i:i + 200000])
    surf.append(s.cpu().numpy())
pts = cand[np.concatenate(surf)]
chunks = [torch.from_numpy(pts[i * N:(i + 1) * N]).to(dev).float().contiguous() for i in range(8)]
L, ctx = pm._L, pm._ctx
f = dict(dtype=torch.float32, device=dev)
streams = pm.side_streams(2)
state = []
ranks = list(pm.RANKS)
for k, st in enumerate(streams):
    with torch.cuda.stream(st):
        vis, ori, conf, mask = torch.empty((V, N), **f), torch.empty((V, N, 2), **f), torch.empty((V, N), **f), torch.empty((V, N), **f)
        bidx = torch.empty((20, N), dtype=torch.int32, device=dev)
        bval = torch.empty((20, N), **f)
        lo, ml = torch.empty((N, 3), **f), torch.empty((N,), **f)
        hc = torch.empty((N,), dtype=torch.bool, device=dev)
        scratch, need = pm._get_scratch(N)
        p = chunks[k]
        sp = _lib.stream_ptr()
        _lib.check(L.mh_forward_prepare(ctx, _lib.ptr(p), N, pm._side, float(pm.conf_threshold), _lib.ptr(vis), _lib.ptr(ori),
                                        _lib.ptr(conf), _lib.ptr(mask), _lib.ptr(scratch), need, sp))
        _lib.check(L.mh_topk_views(ctx, _lib.ptr(vis), _lib.ptr(conf), N, _lib.ptr(bidx), _lib.ptr(bval), sp))
        state.append((p, ori, bidx, bval, scratch, lo, ml, hc))
torch.cuda.synchronize()


def search(k):
    p, ori, bidx, bval, scratch, lo, ml, hc = state[k]
    with torch.cuda.stream(streams[k]):
        _lib.check(L.mh_search_prepared(ctx, _lib.ptr(p), N, pm._side, float(pm.conf_threshold), len(ranks),
                                        ranks[1] - ranks[0], _lib.ptr(ori), _lib.ptr(bidx), _lib.ptr(bval),
                                        _lib.ptr(scratch), _lib.ptr(lo), _lib.ptr(ml), _lib.ptr(hc), None, None, None,
                                        _lib.stream_ptr()))


for i in range(60):
    search(i % 2)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 200
for i in range(K):
    search(i % 2)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("search only, 2 streams: %.4f ms per launch" % (dt / K * 1e3))
for i in range(60):
    with torch.cuda.stream(streams[i % 2]):
        pm.forward(pts[(i % 8) * N:(i % 8 + 1) * N])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    with torch.cuda.stream(streams[i % 2]):
        pm.forward(pts[(i % 8) * N:(i % 8 + 1) * N])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("full iteration, 2 streams: %.4f ms" % (dt / K * 1e3))
