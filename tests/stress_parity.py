#!/usr/bin/env python
"""Randomised HIP-vs-oracle parity sweep (GPU box): many scenes with random camera counts / image sizes / patch sizes /
thresholds / map quantisation / point sets, every result compared bit for bit with the CPU oracle.

    python tests/stress_parity.py --minutes 5 [--seed 0]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (this tool is a checker, like the tests)
from monohair_amd import synth  # noqa: E402
from monohair_amd.camera import camera_records, cameras_from_list  # noqa: E402
from monohair_amd.pmvo import PMVO, depth_offsets  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--minutes", type=float, default=3.0)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--variant", type=int, default=0, help="search-kernel variant (0 = shipped default)")
ap.add_argument("--max-views", type=int, default=70, help="scenes draw 20 .. max-views cameras (above 64: several view chunks\n"
                "of the search, above 256: third level of the view cascade)")
ap.add_argument("--codes", action="store_true", help="maps uploaded as random 8-bit CODES (PMVO.from_u8: the code-gather front "
                "end, mh_project_taps_codes_kernel) against the oracle on the table-decoded maps")
ap.add_argument("--ori-mode", default="", help="adversarial orientation fields for the tap search (continuous maps only): "
                "'mix' draws per scene from: const (one direction per view + tiny noise on a few pixels: near-ties, lists of "
                "1-3 taps), two (two exactly perpendicular directions), axis ((1,0)/(0,1)/(0,0)), fine (a fan of directions "
                "1e-7..1e-4 rad apart: many taps on the flat top of the cosine, losses 0 and below).  There is no NaN mode: NaN pixels "
                "in the orientation MAPS are outside the pinned domain (map files are 8-bit; docs/PARITY.md) -- the reference and "
                "the oracle turn a NaN first tap of a view that does NOT see the point into a NaN loss (NaN x weight 0), the "
                "kernels never gather the patches of such views.  NaN taps of views that do see the point agree: "
                "tests/test_key_reeval_gpu.py")
ap.add_argument("--body", type=int, default=0, help="tap body of the shipped search: 0 by the maps, 1 keys, 2 select (option search_body)")
a = ap.parse_args()
# the oracle's rule state is process-global: whatever this sweep draws per scene is put back when it ends, however it ends
import atexit  # noqa: E402

_RULES0 = (oracle.get_reproject_rule(), oracle.set_sum_block(32))
oracle.set_sum_block(_RULES0[1])
atexit.register(lambda: (oracle.set_reproject_rule(*_RULES0[0]), oracle.set_sum_block(_RULES0[1])))
rng = np.random.default_rng(a.seed)
DEV = "cuda:0"
offs = depth_offsets(90)
t_end = time.time() + a.minutes * 60
n_scene = n_pts = 0
bad = []


def eq(x, y):
    return np.array_equal(x, y, equal_nan=True)


while time.time() < t_end:
    V = int(rng.integers(20, a.max_views))
    big = V > 80                         # many views: small images keep a scene to seconds
    H = int(rng.integers(48, 120 if big else 400))
    W = int(rng.integers(40, 100 if big else 300))
    patch = int(rng.choice([1, 3, 5, 7, 9, 11]))
    thr = float(rng.choice([0.05, 0.1, 0.15, 0.3, 0.6]))
    quant = bool(rng.integers(0, 2))
    rings = int(rng.integers(1, 4))
    scale = float(rng.uniform(0.8, 2.6))
    seed = int(rng.integers(0, 1 << 30))
    scene = synth.make_scene(V, H, W, seed=seed, quantize=quant, rings=rings, scale=scale)
    # perturb the cameras off the ring (translations, principal point) so that projections are generic
    for c in scene["cams"]:
        c["pose"] = (np.array(c["pose"]) + np.pad(rng.normal(0, 0.01, (3, 1)), ((0, 1), (3, 0)))).tolist()
        c["ndc_prj"][2] = float(rng.normal(0, 0.02))
        c["ndc_prj"][3] = float(rng.normal(0, 0.02))
    camd = cameras_from_list(scene["cams"])
    rec = camera_records(camd)
    mode = ""
    if a.codes:
        from monohair_amd.pmvo_utils import map_code_lut

        lut = map_code_lut()
        sc = synth.make_scene_codes(V, H, W, seed=seed, rings=rings, scale=scale)
        k8, c8, m8 = sc["ori_u8"].numpy().copy(), sc["conf_u8"].numpy().copy(), sc["mask_u8"].numpy()
        flip = rng.random(k8.shape) < rng.uniform(0.0, 0.6)           # from clean 2-3-code patches to 49 distinct codes
        k8[flip] = rng.integers(0, 256, size=int(flip.sum())).astype(np.uint8)
        low = rng.random(c8.shape) < rng.uniform(0.0, 0.5)
        c8[low] = rng.integers(0, 120, size=int(low.sum())).astype(np.uint8)
        pm = PMVO.from_u8(camd, sc["depth"].numpy(), k8, c8, m8, device=DEV, image_size=[H, W], patch_size=patch,
                          visible_threshold=1, conf_threshold=thr, records=rec)
        scene = dict(scene, depth=sc["depth"], ori=torch.from_numpy(lut[k8][..., :2].copy()),
                     conf=torch.from_numpy(lut[c8][..., 2].copy()), mask=torch.from_numpy(lut[m8][..., 3].copy()))
    else:
        if a.ori_mode:
            mode = a.ori_mode if a.ori_mode != "mix" else str(rng.choice(["const", "two", "axis", "fine", "plain"]))
            o = scene["ori"].numpy().copy()
            ang = rng.uniform(0, np.pi, size=(V, 1, 1))
            if mode == "const":
                th = ang + (rng.random(o.shape[:3]) < rng.uniform(0.0, 0.2)) * rng.normal(0, 10.0 ** rng.uniform(-7, -2), o.shape[:3])
                o = np.stack([np.cos(th), np.sin(th)], -1).astype(np.float32)
            elif mode == "two":
                pick = rng.random(o.shape[:3]) < 0.5
                c, s_ = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)
                o = np.where(pick[..., None], np.stack([c + 0 * pick, s_ + 0 * pick], -1), np.stack([-s_ + 0 * pick, c + 0 * pick], -1)).astype(np.float32)
            elif mode == "axis":
                k = rng.integers(0, 5, o.shape[:3])
                tab = np.array([[1, 0], [0, 1], [0, 0], [-1, 0], [0.6, 0.8]], np.float32)
                o = tab[k]
            elif mode == "fine":
                th = ang + rng.integers(0, 64, o.shape[:3]) * 10.0 ** rng.uniform(-7.5, -4)
                o = (np.stack([np.cos(th), np.sin(th)], -1) * rng.uniform(0.5, 2.0, o.shape[:3] + (1,))).astype(np.float32)
            elif mode != "plain":
                raise SystemExit("--ori-mode must be one of mix, const, two, axis, fine, plain")
            scene["ori"] = torch.from_numpy(np.ascontiguousarray(o, dtype=np.float32))
        pm = PMVO.from_planes(rec, scene["depth"].to(DEV), scene["ori"].to(DEV), scene["conf"].to(DEV),
                              scene["mask"].to(DEV), device=DEV, patch_size=patch, visible_threshold=1, conf_threshold=thr)
    if a.variant:
        pm.set_option("search_variant", a.variant)
    if a.body:
        pm.set_option("search_body", a.body)
    # the batch rules: mostly the defaults, with MKL's switch to the chain form drawn low enough that
    # batches of a few hundred points have groups on both sides of it; sometimes one form forced, sometimes no trailing columns
    rule = int(rng.choice([0, 0, 0, 0, 1, 2]))
    fma_cols = int(rng.choice([28445, 90 * int(rng.integers(2, 60))]))
    block = int(rng.choice([32, 32, 32, 0]))
    pm.set_option("reproject_rule", rule)
    pm.set_option("reproject_fma_min_cols", fma_cols)
    pm.set_option("sum_block", block)
    oracle.set_reproject_rule({0: "group", 1: "mid", 2: "chain"}[rule], fma_cols)
    oracle.set_sum_block(block)
    views = oracle.Views(rec, scene["depth"].numpy(), scene["ori"].numpy(), scene["conf"].numpy(), scene["mask"].numpy())
    N = 1 if rng.random() < 0.04 else int(rng.integers(1, 400))     # (batches of ONE point: their own sum order)
    cand = synth.candidate_points(res=int(rng.choice([32, 64])), seed=seed % 1000)
    pts = cand[rng.choice(len(cand), N, replace=False)] * rng.uniform(0.9, 1.1)
    for fused in (True, False):
        _, ori, loss, hc = pm.forward(pts, fused=fused)
        _, o_ori, o_loss, o_hc = oracle.forward(views, pts, patch, thr, offs)
        if not (eq(loss.cpu().numpy(), o_loss) and eq(ori.cpu().numpy(), o_ori) and eq(hc.cpu().numpy(), o_hc)):
            bad.append(("forward", fused, V, H, W, patch, thr, quant, seed, N, mode, rule, fma_cols, block))
    surf, _, filt = pm.filter_points(pts)
    unv = pm.compute_unvisible_points(pts)
    o_s, o_f, o_u, _ = oracle.filter_votes(views, pts, patch, thr, 1.0)
    if not (eq(surf.cpu().numpy(), o_s) and eq(filt.cpu().numpy(), o_f) and eq(unv.cpu().numpy(), o_u)):
        bad.append(("filter", V, H, W, patch, thr, quant, seed, N))
    dirs = rng.normal(size=(N, 3)).astype(np.float32)
    pm.Compute_Visible_and_Ori(pts)
    rl, _ = pm.prj_loss_of(pm._points, torch.from_numpy(dirs).to(DEV))
    o_rl, _ = oracle.refine_loss(views, pts, dirs, patch, thr)
    if not eq(rl.cpu().numpy(), o_rl):
        bad.append(("refine_loss", V, H, W, patch, thr, quant, seed, N))
    # the step-by-step methods of the class and the fused refine kernels on the same scene
    base = oracle.topk_views(pm.visible.cpu().numpy(), pm.Conf.cpu().numpy(), 20)[0][0]
    smp, _ = pm.sample_next_3d_pos(pts, base)
    o_smp = oracle.sample_next(views, pts, base, pm.Ori.cpu().numpy(), offs)
    D = pm.compute_reproject_ori(pts, smp)
    o_D = oracle.reproject_ori(views, pts, o_smp)
    l3, i3, h3 = pm.compute_prj_loss(D)
    o = oracle.visible_and_ori(views, pts, patch)
    o_l3, o_i3, o_h3 = oracle.prj_loss(o_D, o["Ori_patch"], o["Conf_patch"], o["visible"], thr)
    if not (eq(smp.cpu().numpy(), o_smp) and eq(D.cpu().numpy(), o_D) and eq(l3.cpu().numpy(), o_l3)
            and eq(i3.cpu().numpy(), o_i3) and eq(h3.cpu().numpy(), o_h3)):
        bad.append(("pieces", V, H, W, patch, thr, quant, seed, N))
    lf = torch.empty((N,), device=DEV)
    from monohair_amd import _lib as L_

    L_.check(pm._L.mh_refine_loss_maps(pm._ctx, L_.ptr(pm._points), L_.ptr(torch.from_numpy(dirs).to(DEV).contiguous()),
                                       0.005, 4.0, N, patch, float(thr), L_.ptr(lf), None, 0, 0, 0, L_.stream_ptr()))
    if not eq(lf.cpu().numpy(), o_rl):
        bad.append(("refine_loss_maps", V, H, W, patch, thr, quant, seed, N))
    n_scene += 1
    n_pts += N
print({"scenes": n_scene, "points": n_pts, "mismatching_cases": len(bad), "first": bad[:5]})
sys.exit(1 if bad else 0)
