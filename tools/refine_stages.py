#!/usr/bin/env python
"""Stage timers of refine() at the headline size (two passes in one process: the first includes every one-time cost,
the second is steady state).  python tools/refine_stages.py"""
import os, sys, tempfile, time, types
os.environ["MH_TIMING"] = "1"
import numpy as np, torch
from scipy.spatial import KDTree
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monohair_amd import synth
from monohair_amd.camera import camera_records, cameras_from_list
from monohair_amd.pmvo import PMVO, filter_negative_points, optimize, refine
dev = torch.device("cuda", 0)
scene = synth.make_scene(60, 1920, 1080, device=dev)
cams = cameras_from_list(scene["cams"])
pm = PMVO.from_planes(camera_records(cams), scene["depth"], scene["ori"], scene["conf"], scene["mask"], device=dev,
                      patch_size=7, visible_threshold=1, conf_threshold=0.15, camera=cams)
rng = np.random.default_rng(1)
b = rng.normal(size=(2000, 3)); b = b / np.linalg.norm(b, axis=1, keepdims=True) * 0.09
scalp = b[b[:, 1] > 0.03] * (0.1 / 0.09)
pm.set_head(KDTree(b), KDTree(scalp), scalp.max(0))
cand = synth.candidate_points(res=256, seed=0)
tmp = tempfile.mkdtemp()
args = types.SimpleNamespace(device=str(dev), output_path=tmp, save_root=tmp + "/optimize", save_path=tmp + "/refine",
                             PMVO=types.SimpleNamespace(visible_threshold=1), data=types.SimpleNamespace(root=tmp))
os.makedirs(args.save_path, exist_ok=True)
s_idx, s_pts, f_idx = filter_negative_points(cand, pm, args)
sp, so, ml, hc = optimize(s_pts, pm, args)
for rep in range(2):
    print("---- refine pass", rep, file=sys.stderr)
    t = time.perf_counter()
    refine(sp.copy(), so.copy(), ml.copy(), pm, cand[:len(f_idx)][f_idx].astype(np.float32), args, infer_inner=False,
           threshold=0.025, return_dense=False)
    torch.cuda.synchronize()
    print("refine total %.1f ms" % ((time.perf_counter() - t) * 1e3), file=sys.stderr)
