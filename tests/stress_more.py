#!/usr/bin/env python
"""Randomised sweeps of the other kernels against their checkers (GPU box):
  k-NN grid search vs scipy.spatial.KDTree, depth rasteriser / Gabor bank (all variants) / medoid / voxel fit vs the
  CPU oracle.  Every comparison is for exact equality.
    python tests/stress_more.py --minutes 5 [--seed 0]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
from scipy.spatial import KDTree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (this tool is a checker, like the tests)
from monohair_amd import synth  # noqa: E402
from monohair_amd.camera import camera_records, cameras_from_list  # noqa: E402
from monohair_amd.gabor import calOrientationGabor, gabor_bank  # noqa: E402
from monohair_amd.pmvo_utils import GridKNN, compute_points_similarity, voxel_fit  # noqa: E402
from monohair_amd import _lib  # noqa: E402
from monohair_amd.render import DepthRenderer, StrandRenderer, strand_line_buffers  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--minutes", type=float, default=3.0)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
DEV = "cuda:0"
t_end = time.time() + a.minutes * 60
count = {"knn": 0, "raster": 0, "strands": 0, "gabor": 0, "medoid": 0, "voxel_fit": 0, "trace": 0, "votes": 0}
bad = []
gabs = {v: calOrientationGabor(device=DEV, variant=v) for v in ("mfma2", "valu")}
bank = gabor_bank()


def cloud(n):
    kind = rng.integers(0, 4)
    if kind == 0:
        return rng.random((n, 3))
    if kind == 1:                                                     # shell (what refine sees)
        p = rng.normal(size=(n, 3))
        return p / np.linalg.norm(p, axis=1, keepdims=True) * (0.12 + rng.normal(0, 0.002, (n, 1)))
    if kind == 2:                                                     # clusters of very different density
        return np.concatenate([rng.normal(0, 0.005, (n // 2, 3)), rng.normal(0.5, 0.2, (n - n // 2, 3))])
    return rng.normal(0, 1, (n, 3)) * np.array([1.0, 0.05, 0.3])      # anisotropic slab


while time.time() < t_end:
    # ---- k-NN
    n = int(rng.integers(50, 60000))
    pts = cloud(n).astype(np.float32)
    k = int(rng.choice([1, 7, 32, 100]))
    q = np.concatenate([pts[rng.choice(n, min(n, 300), replace=False)],
                        (pts[:100] + rng.normal(0, 0.01, (min(n, 100), 3))).astype(np.float32)])
    got = GridKNN(pts, k_hint=k, device=DEV).query(q, k).cpu().numpy()
    kk = min(k, n)
    d, ref = KDTree(data=pts).query(q, kk)
    ref = np.asarray(ref).reshape(len(q), kk)
    if not np.array_equal(got, ref):
        dg = np.linalg.norm(pts[got].astype(np.float64) - q[:, None].astype(np.float64), axis=-1)
        if not np.allclose(dg, np.asarray(d).reshape(len(q), kk), rtol=0, atol=0):    # only exact distance ties may differ
            bad.append(("knn", n, k))
    count["knn"] += 1
    # ---- rasteriser
    H, W = int(rng.integers(20, 300)), int(rng.integers(20, 300))
    cams = synth.make_cameras(20, H, W, scale=float(rng.uniform(0.6, 2.5)), rings=int(rng.integers(1, 3)))
    rec = camera_records(cameras_from_list(cams))[int(rng.integers(0, 20))]
    nv = int(rng.integers(3, 3000))
    verts = (rng.normal(0, 0.1, (nv, 3)) * rng.uniform(0.2, 3)).astype(np.float32)
    faces = rng.integers(0, nv, (int(rng.integers(1, 6000)), 3)).astype(np.int32)
    pc = float(rng.choice([0.0, 0.5, 0.25]))
    want, _ = oracle.render_depth(rec, verts, faces, H, W, pc)
    got = DepthRenderer([(verts, faces)], DEV).render(rec, H, W, pc).cpu().numpy()
    if not np.array_equal(got, want):
        bad.append(("raster", H, W, nv, len(faces), pc))
    count["raster"] += 1
    # ---- strand-segment renderer: random polylines (many sub-pixel segments, coordinates on pixel borders after the
    # coarse snapping), both line rules, 1..3 pixel wide, 4..8 sub-pixel bits
    strands = []
    for _ in range(int(rng.integers(1, 60))):
        m = int(rng.integers(2, 40))
        p0 = rng.normal(0, 0.08, 3)
        strands.append((p0 + np.cumsum(rng.normal(0, rng.uniform(0.0005, 0.01), (m, 3)), 0)).astype(np.float32))
    lp, lt = strand_line_buffers(strands)
    sr = StrandRenderer(strands, verts, faces[: int(rng.integers(0, 200))], DEV)
    rule, width, sbits = int(rng.integers(0, 2)), int(rng.integers(1, 4)), int(rng.integers(4, 9))
    copt, dopt = int(rng.integers(0, 4)), int(rng.integers(0, 3))
    oracle.set_subpixel_bits(sbits)
    _lib.check(_lib.lib().mh_ctx_set_option(sr._ctx, b"raster_subpixel_bits", sbits))
    try:
        want, _, _ = oracle.render_strands(rec, sr.verts.cpu().numpy(), sr.faces.cpu().numpy(), lp, lt, H, W, pc, width, copt,
                                           dopt, 0.25, line_rule=rule)
        got = sr.render(rec, H, W, copt, dopt, 0.25, pixel_center=pc, line_width=width, line_rule=rule).cpu().numpy()
    finally:
        oracle.set_subpixel_bits(8)
        _lib.lib().mh_ctx_set_option(sr._ctx, b"raster_subpixel_bits", 8)
    if not np.array_equal(got, want):
        bad.append(("strands", H, W, len(lp), rule, width, sbits, copt, dopt, pc))
    count["strands"] += 1
    # ---- Gabor bank, three kernels
    H, W = int(rng.integers(5, 90)), int(rng.integers(5, 120))
    img = (rng.normal(size=(H, W)) * rng.uniform(0.01, 3)).astype(np.float32)
    if rng.integers(0, 3) == 0:
        img[:, : W // 2] = 0.0                                        # flat regions: zero responses, argmax ties
    o_idx, o_conf, o_var = oracle.gabor_bank(bank, img)
    for v, g in gabs.items():
        idx, conf, var = g.filter_index(torch.from_numpy(img).to(DEV))
        if not (np.array_equal(idx.cpu().numpy(), o_idx) and np.array_equal(conf.cpu().numpy(), o_conf, equal_nan=True)
                and np.array_equal(var.cpu().numpy(), o_var)):
            bad.append(("gabor", v, H, W))
    count["gabor"] += 1
    # ---- medoid
    G, K = int(rng.integers(1, 400)), int(rng.integers(1, 130))
    ori = rng.normal(size=(G, K, 3)).astype(np.float32)
    if rng.integers(0, 3) == 0:
        ori[:, K // 2:] = ori[:, : K - K // 2]                        # duplicates -> ties
    got = compute_points_similarity(torch.from_numpy(ori).to(DEV)).cpu().numpy()
    want, _ = oracle.medoid_dense(ori)
    if not np.array_equal(got, want):
        bad.append(("medoid", G, K))
    count["medoid"] += 1
    # ---- voxel fit
    n = int(rng.integers(1, 20000))
    p = cloud(n) * 0.2
    o = rng.normal(size=(n, 3)).astype(np.float32)
    res = voxel_fit(p.copy(), o.copy(), DEV)
    occ, ori_d = oracle.voxel_fit(p.copy(), o.copy(), [-0.32, -0.32, -0.24], 0.005 / 2, [256, 256, 192])
    if not (np.array_equal(res["occ"], occ) and np.array_equal(res["ori_dense"].astype(np.float32), ori_d.astype(np.float32))):
        bad.append(("voxel_fit", n))
    count["voxel_fit"] += 1
    # ---- strand tracing on a random smooth-ish volume
    Z, Hh, Ww = int(rng.integers(8, 40)), int(rng.integers(8, 48)), int(rng.integers(8, 48))
    occ = (rng.random((Z, Hh, Ww)) < rng.uniform(0.05, 0.6)).astype(np.float32)
    base = rng.normal(size=3)
    ori = (base + rng.normal(0, rng.uniform(0.05, 1.0), (Z, Hh, Ww, 3))).astype(np.float32)
    ori /= np.maximum(np.linalg.norm(ori, axis=-1, keepdims=True), 1e-6)
    ori *= occ[..., None]
    from monohair_amd.hairgrow import HairGrowing

    hg = HairGrowing(None, None, device=DEV, occ=occ[..., None], ori=ori)
    vol = oracle.Volume(occ, ori)
    nz = np.argwhere(vol.vox[..., 3] != 0)
    if len(nz):
        seeds = (nz[rng.choice(len(nz), min(len(nz), 400), replace=False)][:, ::-1] + rng.random((min(len(nz), 400), 3))).astype(np.float32)
        thr = float(rng.choice([0.5, 0.8, 0.95]))
        out, first, ln = hg._trace_seeds(torch.from_numpy(seeds).to(DEV), thr)
        o_out, o_first, o_ln = oracle.trace_seeds(vol, seeds, thr)
        ok = np.array_equal(first.cpu().numpy(), o_first) and np.array_equal(ln.cpu().numpy(), o_ln)
        if ok:
            out = out.cpu().numpy()
            ok = all(np.array_equal(out[i, o_first[i]:o_first[i] + o_ln[i]], o_out[i, o_first[i]:o_first[i] + o_ln[i]])
                     for i in range(len(seeds)))
        nrm = rng.normal(size=seeds.shape).astype(np.float32)
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        sp, sl = hg._trace_scalp(torch.from_numpy(seeds).to(DEV), torch.from_numpy(nrm).to(DEV), thr)
        o_sp, o_sl = oracle.trace_scalp(vol, seeds, nrm, thr)
        ok = ok and np.array_equal(sl.cpu().numpy(), o_sl)
        if ok:
            sp = sp.cpu().numpy()
            ok = all(np.array_equal(sp[i, :o_sl[i]], o_sp[i, :o_sl[i]]) for i in range(len(o_sl)))
        if not ok:
            bad.append(("trace", Z, Hh, Ww, thr))
    count["trace"] += 1
    # ---- the votes of LARGE launches (round 6: mh_filter_rows_kernel, lane = point, rows taken in grid-cell order, the trailing
    # rows of every batch by the wave-per-point kernel) against the oracle batch by batch: random views / patch / batch length /
    # slice of a longer run, continuous and 8-bit maps
    if count["trace"] % 3 == 0:
        import ctypes

        from monohair_amd.pmvo import PMVO
        from monohair_amd.pmvo_utils import spatial_order

        Vv, Hv, Wv = int(rng.integers(20, 70)), int(rng.integers(48, 200)), int(rng.integers(40, 160))
        patch = int(rng.choice([1, 3, 5, 7, 9]))
        thr = float(rng.choice([0.05, 0.15, 0.4]))
        scene = synth.make_scene(Vv, Hv, Wv, seed=int(rng.integers(0, 1 << 30)), quantize=bool(rng.integers(0, 2)),
                                 rings=int(rng.integers(1, 3)), scale=float(rng.uniform(0.9, 2.4)))
        recs = camera_records(cameras_from_list(scene["cams"]))
        pmv = PMVO.from_planes(recs, scene["depth"].to(DEV), scene["ori"].to(DEV), scene["conf"].to(DEV), scene["mask"].to(DEV),
                               device=DEV, patch_size=patch, visible_threshold=1, conf_threshold=thr)
        views = oracle.Views(recs, scene["depth"].numpy(), scene["ori"].numpy(), scene["conf"].numpy(), scene["mask"].numpy())
        total = int(rng.integers(4200, 15000))
        cand = synth.candidate_points(res=48, seed=int(rng.integers(0, 1000)))
        P = (cand[rng.choice(len(cand), total, replace=total > len(cand))] * rng.uniform(0.9, 1.1, (total, 1))).astype(np.float32)
        batch = int(rng.choice([0, int(rng.integers(1, total + 10)), int(rng.integers(100, 3000))]))
        row0 = 0 if batch == 0 else int(rng.integers(0, total - 4096))
        nrow = total if batch == 0 else int(rng.integers(4096, total - row0 + 1))
        dP = torch.from_numpy(P).to(DEV)
        sub = dP[row0:row0 + nrow].contiguous()
        order = spatial_order(sub) if rng.integers(0, 2) else None
        outs = [torch.full((nrow,), 9, dtype=torch.uint8, device=DEV) for _ in range(4)]
        vt = float(rng.choice([1.0, 0.5]))
        _lib.check(_lib.lib().mh_filter_points_ordered(pmv._ctx, _lib.ptr(sub), nrow, pmv._side, thr, vt,
                                                       *[_lib.ptr(o) for o in outs], batch, row0, total if batch else 0,
                                                       _lib.ptr(order), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   "mh_filter_points_ordered")
        got = [o.cpu().numpy().astype(bool) for o in outs]
        bsz, tot = (batch, total) if batch else (nrow, nrow)
        base = row0 if batch else 0
        want = [np.zeros(nrow, bool) for _ in range(4)]
        for s0 in range(base // bsz * bsz, base + nrow, bsz):
            e0 = min(s0 + bsz, tot)
            src = P if batch else P[row0:row0 + nrow]
            w = oracle.filter_votes(views, src[s0:e0], patch, thr, vt)
            a0, a1 = max(s0, base), min(e0, base + nrow)
            for kq in range(4):
                want[kq][a0 - base:a1 - base] = w[kq][a0 - s0:a1 - s0]
        if not all(np.array_equal(g_, w_) for g_, w_ in zip(got, want)):
            bad.append(("votes", Vv, Hv, Wv, patch, thr, total, batch, row0, nrow, order is not None))
        count["votes"] += 1
        del pmv
print({"rounds": count, "mismatching_cases": len(bad), "first": bad[:6]})
sys.exit(1 if bad else 0)
