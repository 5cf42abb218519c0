#!/usr/bin/env python
"""Where do the 47 ms of optimize() go on the bench scene? (GPU box)"""
import os, sys, time, types, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monohair_amd import synth
from monohair_amd.camera import camera_records, cameras_from_list
from monohair_amd.pmvo import PMVO, optimize
import monohair_amd.pmvo as P

dev = torch.device("cuda", 0)
scene = synth.make_scene(60, 1920, 1080, device=dev, seed=0)
cams = cameras_from_list(scene["cams"])
pm = PMVO.from_planes(camera_records(cams), scene["depth"], scene["ori"], scene["conf"], scene["mask"], device=dev,
                      patch_size=7, visible_threshold=1, conf_threshold=0.15, camera=cams)
cand = synth.candidate_points(res=256, seed=0)
surf = np.concatenate([pm.filter_points(cand[i:i + 200000])[0].cpu().numpy() for i in range(0, len(cand), 200000)])
pts = cand[surf].astype(np.float32)
tmp = tempfile.mkdtemp()
args = types.SimpleNamespace(save_root=tmp + "/optimize")
for k in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    optimize(pts, pm, args)
    torch.cuda.synchronize(); print("optimize() %.1f ms" % ((time.perf_counter() - t0) * 1e3))
# parts
chunks = [pts[i * 5000:(i + 1) * 5000] for i in range(len(pts) // 5000 + 1)]
streams = pm.side_streams(3)
for k in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = []
    for i, c in enumerate(chunks):
        with torch.cuda.stream(streams[i % 3]):
            _, o, l, h = pm.forward(c)
            outs.append((o, l, h))
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    o = torch.cat([x[0] for x in outs]); l = torch.cat([x[1] for x in outs]); h = torch.cat([x[2] for x in outs])
    on, ln, hn = o.cpu().numpy(), l.cpu().numpy(), h.cpu().numpy()
    t3 = time.perf_counter()
    np.save(tmp + "/a.npy", pts); np.save(tmp + "/b.npy", on); np.save(tmp + "/c.npy", ln); np.save(tmp + "/d.npy", hn)
    t4 = time.perf_counter()
    print("enqueue %.1f ms, gpu done %.1f ms (%.3f ms/chunk), cat+D2H %.1f ms, np.save %.1f ms" % (
        (t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) * 1e3 / len(chunks), (t3 - t2) * 1e3, (t4 - t3) * 1e3))
