#!/usr/bin/env python
"""Time the full exterior pass (filter -> optimize -> refine -> volume) on the synthetic workload, stage by stage.
   python tools/full_pass.py [--views 60 --height 1920 --width 1080 --volume 256 --patch 7]"""
import argparse
import os
import sys
import tempfile
import time
import types

import numpy as np
import torch
from scipy.spatial import KDTree

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monohair_amd import synth  # noqa: E402
from monohair_amd.camera import camera_records, cameras_from_list  # noqa: E402
from monohair_amd.pmvo import PMVO, filter_negative_points, optimize, refine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=60)
ap.add_argument("--height", type=int, default=1920)
ap.add_argument("--width", type=int, default=1080)
ap.add_argument("--volume", type=int, default=256)
ap.add_argument("--patch", type=int, default=7)
ap.add_argument("--quantize", action="store_true")
ap.add_argument("--profile", action="store_true", help="cProfile the refine stage (host-side view)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
T = {}


def tic(name, t0):
    torch.cuda.synchronize()
    T[name] = round(time.perf_counter() - t0, 3)


t0 = time.perf_counter()
scene = synth.make_scene(a.views, a.height, a.width, device=dev, quantize=a.quantize)
cams = cameras_from_list(scene["cams"])
pm = PMVO.from_planes(camera_records(cams), scene["depth"], scene["ori"], scene["conf"], scene["mask"], device=dev,
                      patch_size=a.patch, visible_threshold=1, conf_threshold=0.15, camera=cams)
rng = np.random.default_rng(1)
b = rng.normal(size=(2000, 3))
b = b / np.linalg.norm(b, axis=1, keepdims=True) * 0.09
scalp = b[b[:, 1] > 0.03] * (0.1 / 0.09)
pm.set_head(KDTree(b), KDTree(scalp), scalp.max(0))
tic("scene+pack_s", t0)
cand = synth.candidate_points(res=a.volume, seed=0)
tmp = tempfile.mkdtemp()
args = types.SimpleNamespace(device=str(dev), output_path=tmp, save_root=tmp + "/optimize", save_path=tmp + "/refine",
                             PMVO=types.SimpleNamespace(visible_threshold=1), data=types.SimpleNamespace(root=tmp))
os.makedirs(args.save_path, exist_ok=True)
t0 = time.perf_counter()
s_idx, s_pts, f_idx = filter_negative_points(cand, pm, args)
tic("filter_s", t0)
t0 = time.perf_counter()
sp, so, ml, hc = optimize(s_pts, pm, args)
tic("optimize_s", t0)
if a.profile:
    import cProfile
    import pstats

    pr = cProfile.Profile()
    pr.enable()
t0 = time.perf_counter()
refine(sp.copy(), so.copy(), ml.copy(), pm, cand[:len(f_idx)][f_idx].astype(np.float32), args, infer_inner=False,
       threshold=0.025, return_dense=False)          # as PMVO.py does: Ori3D.mat / Occ3D.mat written from the voxel list
tic("refine+volume_s", t0)
if a.profile:
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
# --- strand tracing on the fitted volume (scalp roots + two voxel-seeded rounds)
from monohair_amd.hairgrow import HairGrowing  # noqa: E402

from monohair_amd.pmvo_utils import get_ground_truth_3D_occ, get_ground_truth_3D_ori  # noqa: E402

t0 = time.perf_counter()
occ_zyx = get_ground_truth_3D_occ(args.save_path + "/Occ3D.mat")       # what HairGrow.py reads (HairGrow.py:816-824)
ori_zyx = get_ground_truth_3D_ori(args.save_path + "/Ori3D.mat")
T["voxels"] = int(occ_zyx.sum())
tic("load_mat_s", t0)
t0 = time.perf_counter()
hg = HairGrowing(None, None, device=str(dev), occ=occ_zyx, ori=ori_zyx)
rs = np.random.default_rng(2)
nrm = rs.normal(size=(60000, 3))
nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
nrm[:, 1] = -np.abs(nrm[:, 1])
nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
centre = np.array([128.0, 128.0, 96.0])
sp = torch.from_numpy((centre + nrm * 40.0).astype(np.float32))         # 10 cm scalp sphere in voxel units
torch.manual_seed(0)
if a.profile:
    pr = cProfile.Profile()
    pr.enable()
strands, num_root = hg.GenerateGuideStrandFromScalp(sp, torch.from_numpy(nrm.astype(np.float32)), None, 0.85)
tic("trace_strands_s", t0)
if a.profile:
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
T["strands"] = len(strands)
T["strand_points"] = int(sum(s.shape[0] for s in strands))
T["num_root"] = num_root
print({"candidates": len(cand), "surface": int(s_idx.sum()), "shell": int(f_idx.sum()),
       "iterations": len(s_pts) // 5000 + 1, **T})
