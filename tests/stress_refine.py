#!/usr/bin/env python
"""Randomised sweep of the `refine` driver (GPU box) against the CPU oracle: the Gauss-Seidel smoothing loop over 5000-point
chunks (PMVO.py:602-643), the shell points (:655-691) and the voxel fit (:695-764), on random scenes and point sets whose
sizes sit on every edge of the chunking: fewer points than the 100 neighbours, exact multiples of 5000 (the empty trailing
chunk of `step = N // 5000 + 1`), one point more / less, several ragged chunks; NaN orientations and losses as `optimize`
leaves them for points no view sees; both single-rank forms of the loop.  Every comparison is for exact equality.

    python tests/stress_refine.py --minutes 5 [--seed 0]
"""
import argparse
import os
import shutil
import sys
import tempfile
import time
import types

import numpy as np
import torch
from scipy.spatial import KDTree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (this tool is a checker, like the tests)
from monohair_amd import synth  # noqa: E402
from monohair_amd.camera import camera_records, cameras_from_list  # noqa: E402
from monohair_amd.pmvo import PMVO, refine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--minutes", type=float, default=3.0)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
DEV = "cuda:0"
t_end = time.time() + a.minutes * 60
bad, n_case, n_pts = [], 0, 0
SIZES = [1, 2, 37, 99, 100, 101, 640, 4999, 5000, 5001, 9999, 10000, 10001, 15000]


def eq(x, y):
    return np.array_equal(x, y, equal_nan=True)


rngb = np.random.default_rng(123)
bust = rngb.normal(size=(800, 3))
bust = bust / np.linalg.norm(bust, axis=1, keepdims=True) * 0.09
scalp = bust[bust[:, 1] > 0.03] * (0.1 / 0.09)
cand_all = {r: synth.candidate_points(res=r, seed=1) for r in (64, 96)}

while time.time() < t_end:
    V = int(rng.integers(20, 40))
    H, W = int(rng.integers(64, 260)), int(rng.integers(48, 200))
    patch = int(rng.choice([3, 5, 7]))
    thr = float(rng.choice([0.1, 0.15, 0.3]))
    seed = int(rng.integers(0, 1 << 30))
    scene = synth.make_scene(V, H, W, seed=seed, quantize=bool(rng.integers(0, 2)), scale=float(rng.uniform(1.2, 2.2)))
    camd = cameras_from_list(scene["cams"])
    rec = camera_records(camd)
    pm = PMVO.from_planes(rec, scene["depth"].to(DEV), scene["ori"].to(DEV), scene["conf"].to(DEV), scene["mask"].to(DEV),
                          device=DEV, patch_size=patch, visible_threshold=1, conf_threshold=thr)
    pm.set_head(KDTree(data=bust), KDTree(data=scalp), np.max(scalp, axis=0))
    views = oracle.Views(rec, scene["depth"].numpy(), scene["ori"].numpy(), scene["conf"].numpy(), scene["mask"].numpy())
    cand = cand_all[int(rng.choice([64, 96]))]
    N = int(rng.choice(SIZES)) if rng.random() < 0.7 else int(rng.integers(1, 16000))
    N = min(N, len(cand))
    pts = (cand[rng.choice(len(cand), N, replace=False)] * rng.uniform(0.97, 1.03)).astype(np.float32)
    # orientations: the meridian tangent field + noise, a fraction random, sign flips, a few NaN rows (as optimize leaves them)
    n = pts / np.linalg.norm(pts, axis=1, keepdims=True)
    t = -np.array([0, 1.0, 0])[None] + n[:, 1:2] * n
    t = t / np.maximum(np.linalg.norm(t, axis=1, keepdims=True), 1e-6) + rng.normal(0, rng.uniform(0.02, 0.5), (N, 3))
    wild = rng.random(N) < rng.uniform(0, 0.3)
    t[wild] = rng.normal(size=(int(wild.sum()), 3))
    t *= rng.choice([-1.0, 1.0], size=(N, 1))
    ori = (t / np.linalg.norm(t, axis=1, keepdims=True)).astype(np.float32)
    loss = rng.uniform(0, 0.1, N).astype(np.float32)
    nanrow = rng.random(N) < rng.choice([0.0, 0.01])
    ori[nanrow] = np.nan
    loss[nanrow] = np.nan
    F = int(rng.choice([0, 1, 50, 3000, 7000]))
    shell = (cand[rng.choice(len(cand), min(F, len(cand)), replace=False)] * rng.uniform(0.9, 0.99)) if F else np.zeros((0, 3))
    threshold = float(rng.choice([0.001, 0.05, 0.2]))
    form = "1" if rng.random() < 0.5 else "0"
    os.environ["MH_REFINE_CHAIN"] = form
    tmp = tempfile.mkdtemp(prefix="mh_stress_refine_")
    args = types.SimpleNamespace(device=DEV, output_path=tmp, save_root=os.path.join(tmp, "optimize"),
                                 save_path=os.path.join(tmp, "refine"), PMVO=types.SimpleNamespace(visible_threshold=1.0),
                                 data=types.SimpleNamespace(root=tmp))
    os.makedirs(args.save_path, exist_ok=True)
    case = (V, H, W, patch, thr, seed, N, F, threshold, form)
    try:
        occ, vol = refine(pts.copy(), ori.copy(), loss.copy(), pm, shell.copy(), args, infer_inner=False, threshold=threshold,
                          genrate_ori_only=False)
        got = {k: np.load(os.path.join(tmp, "refine", k + ".npy")) for k in
               ("select_o", "min_loss", "filter_unvisible", "filter_unvisible_ori")}
        w_o, w_l = ori.copy(), loss.copy()
        oracle.refine_loop(views, pts, w_o, w_l, patch, thr, 1.0, KDTree(data=scalp), np.max(scalp, axis=0))
        if not (eq(got["select_o"], w_o) and eq(got["min_loss"], w_l)):
            bad.append(("loop",) + case)
        keep = np.where(w_l < threshold)[0]
        if len(keep) and len(shell):
            kp, ko = oracle.shell_orientations(views, pts[keep], w_o[keep], shell, patch, thr, 1.0, KDTree(data=scalp),
                                               np.max(scalp, axis=0))
        else:
            kp, ko = np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32)
        if not (eq(got["filter_unvisible"], kp) and eq(got["filter_unvisible_ori"], ko)):
            bad.append(("shell",) + case)
        sp, so = np.concatenate([pts[keep], kp], 0), np.concatenate([w_o[keep], ko], 0)
        if len(sp):
            o_occ, o_vol = oracle.voxel_fit(sp.copy(), so.copy(), [-0.32, -0.32, -0.24], 0.005 / 2, [256, 256, 192])
            if not (eq(occ, o_occ) and eq(vol, o_vol)):
                bad.append(("volume",) + case)
    except Exception as e:          # a crash is a finding too
        bad.append(("raised %r" % (e,),) + case)
    shutil.rmtree(tmp, ignore_errors=True)
    n_case += 1
    n_pts += N
print({"cases": n_case, "points": n_pts, "mismatching_cases": len(bad), "first": bad[:5]})
sys.exit(1 if bad else 0)
