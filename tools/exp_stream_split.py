"""Experiment (GPU box, round 5): the front end of an iteration (projection + tap lists, base-view ranking, launch order) on a
LOW-priority stream and the search kernels back to back on a HIGH-priority stream, against the shipped arrangement (whole
iterations rotating over three equal streams).  Three slots of buffers; events order front end k -> search k -> front end k+3.
    python tools/exp_stream_split.py [--codes]"""
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
codes = "--codes" in sys.argv
if codes:
    sc = synth.make_scene_codes(V, H, W, seed=0)
    cams = cameras_from_list(sc["cams"])
    pm = PMVO.from_u8(cams, sc["depth"].numpy(), sc["ori_u8"].numpy(), sc["conf_u8"].numpy(), sc["mask_u8"].numpy(), device=dev,
                      image_size=[H, W], patch_size=7, visible_threshold=1, conf_threshold=0.15)
else:
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
NCH = min(48, len(pts) // N)
chunks = [torch.from_numpy(pts[i * N:(i + 1) * N]).to(dev).float().contiguous() for i in range(NCH)]
L, ctx = pm._L, pm._ctx
f = dict(dtype=torch.float32, device=dev)
lo_p, hi_p = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("stream priority range (lowest, highest):", lo_p, hi_p)
ranks = list(pm.RANKS)
K = 300


def make_slots(front, nslots=3):
    slots = []
    for k in range(nslots):
        with torch.cuda.stream(front):
            slot = dict(vis=torch.empty((V, N), **f), ori=torch.empty((V, N, 2), **f), conf=torch.empty((V, N), **f),
                        mask=torch.empty((V, N), **f), bidx=torch.empty((20, N), dtype=torch.int32, device=dev),
                        bval=torch.empty((20, N), **f), lo=torch.empty((N, 3), **f), ml=torch.empty((N,), **f),
                        hc=torch.empty((N,), dtype=torch.bool, device=dev), ready=torch.cuda.Event(), done=torch.cuda.Event())
            slot["scratch"], slot["need"] = pm._get_scratch(N)
            slot["scratch"] = slot["scratch"].clone()          # its own buffer (the cache is per stream)
        slots.append(slot)
    return slots


def split_run(front, search, slots, iters):
    for i in range(iters):
        s = slots[i % len(slots)]
        p = chunks[i % NCH]
        with torch.cuda.stream(front):
            sp = _lib.stream_ptr()
            front.wait_event(s["done"])
            _lib.check(L.mh_forward_prepare(ctx, _lib.ptr(p), N, pm._side, float(pm.conf_threshold), _lib.ptr(s["vis"]),
                                            _lib.ptr(s["ori"]), _lib.ptr(s["conf"]), _lib.ptr(s["mask"]), _lib.ptr(s["scratch"]),
                                            s["need"], sp))
            _lib.check(L.mh_topk_views(ctx, _lib.ptr(s["vis"]), _lib.ptr(s["conf"]), N, _lib.ptr(s["bidx"]), _lib.ptr(s["bval"]), sp))
            pm.set_option("search_variant", 109 if codes else 9)
            _lib.check(L.mh_search_prepared(ctx, _lib.ptr(p), N, pm._side, float(pm.conf_threshold), len(ranks), ranks[1] - ranks[0],
                                            _lib.ptr(s["ori"]), _lib.ptr(s["bidx"]), _lib.ptr(s["bval"]), _lib.ptr(s["scratch"]),
                                            _lib.ptr(s["lo"]), _lib.ptr(s["ml"]), _lib.ptr(s["hc"]), None, None, None, sp))
            s["ready"].record(front)
        with torch.cuda.stream(search):
            search.wait_event(s["ready"])
            pm.set_option("search_variant", 110 if codes else 10)
            _lib.check(L.mh_search_prepared(ctx, _lib.ptr(p), N, pm._side, float(pm.conf_threshold), len(ranks), ranks[1] - ranks[0],
                                            _lib.ptr(s["ori"]), _lib.ptr(s["bidx"]), _lib.ptr(s["bval"]), _lib.ptr(s["scratch"]),
                                            _lib.ptr(s["lo"]), _lib.ptr(s["ml"]), _lib.ptr(s["hc"]), None, None, None,
                                            _lib.stream_ptr()))
            s["done"].record(search)
    pm.set_option("search_variant", 0)


def timed(fn):
    fn(60)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(K)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


for name, fp, sp_ in (("equal priorities", 0, 0), ("front low / search high", lo_p, hi_p)):
    front, search = torch.cuda.Stream(dev, priority=fp), torch.cuda.Stream(dev, priority=sp_)
    for nslots in (2, 3, 4):
        slots = make_slots(front, nslots)
        ms = timed(lambda n: split_run(front, search, slots, n))
        print("split streams, %-24s %d slots: %.4f ms per iteration = %.1f it/s" % (name, nslots, ms, 1e3 / ms))


def paired_run(fronts, searches, slots, iters):
    """one (front, search) PAIR of streams per slot: the searches of different slots may overlap (tails filled)"""
    for i in range(iters):
        k = i % len(slots)
        s, front, search = slots[k], fronts[k], searches[k]
        p = chunks[i % NCH]
        with torch.cuda.stream(front):
            sp = _lib.stream_ptr()
            front.wait_event(s["done"])
            _lib.check(L.mh_forward_prepare(ctx, _lib.ptr(p), N, pm._side, float(pm.conf_threshold), _lib.ptr(s["vis"]),
                                            _lib.ptr(s["ori"]), _lib.ptr(s["conf"]), _lib.ptr(s["mask"]), _lib.ptr(s["scratch"]),
                                            s["need"], sp))
            _lib.check(L.mh_topk_views(ctx, _lib.ptr(s["vis"]), _lib.ptr(s["conf"]), N, _lib.ptr(s["bidx"]), _lib.ptr(s["bval"]), sp))
            pm.set_option("search_variant", 109 if codes else 9)
            _lib.check(L.mh_search_prepared(ctx, _lib.ptr(p), N, pm._side, float(pm.conf_threshold), len(ranks), ranks[1] - ranks[0],
                                            _lib.ptr(s["ori"]), _lib.ptr(s["bidx"]), _lib.ptr(s["bval"]), _lib.ptr(s["scratch"]),
                                            _lib.ptr(s["lo"]), _lib.ptr(s["ml"]), _lib.ptr(s["hc"]), None, None, None, sp))
            s["ready"].record(front)
        with torch.cuda.stream(search):
            search.wait_event(s["ready"])
            pm.set_option("search_variant", 110 if codes else 10)
            _lib.check(L.mh_search_prepared(ctx, _lib.ptr(p), N, pm._side, float(pm.conf_threshold), len(ranks), ranks[1] - ranks[0],
                                            _lib.ptr(s["ori"]), _lib.ptr(s["bidx"]), _lib.ptr(s["bval"]), _lib.ptr(s["scratch"]),
                                            _lib.ptr(s["lo"]), _lib.ptr(s["ml"]), _lib.ptr(s["hc"]), None, None, None,
                                            _lib.stream_ptr()))
            s["done"].record(search)
    pm.set_option("search_variant", 0)


for name, fp, sp_ in (("equal priorities", 0, 0), ("front low / search high", lo_p, hi_p)):
    for nslots in (2, 3, 4):
        fronts = [torch.cuda.Stream(dev, priority=fp) for _ in range(nslots)]
        searches = [torch.cuda.Stream(dev, priority=sp_) for _ in range(nslots)]
        slots = make_slots(fronts[0], nslots)
        ms = timed(lambda n: paired_run(fronts, searches, slots, n))
        print("paired streams per slot, %-24s %d slots: %.4f ms per iteration = %.1f it/s" % (name, nslots, ms, 1e3 / ms))

streams = pm.side_streams(3)


def rotate(iters):
    for i in range(iters):
        with torch.cuda.stream(streams[i % 3]):
            pm.forward(chunks[i % NCH])


ms = timed(rotate)
print("shipped: whole iterations rotating over 3 streams (device-resident chunks): %.4f ms = %.1f it/s" % (ms, 1e3 / ms))
